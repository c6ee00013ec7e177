#!/usr/bin/env python3
"""Time compute_mmd forward + backward (BASELINE configs[3] shape, reduced batch) on one GPU.
usage: python tools/time_mmd.py [batch] [len] [dim] [dyadic] [linear|rbf]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
A = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = int(sys.argv[2]) if len(sys.argv) > 2 else 64
D = int(sys.argv[3]) if len(sys.argv) > 3 else 4
d = int(sys.argv[4]) if len(sys.argv) > 4 else 2
kind = sys.argv[5] if len(sys.argv) > 5 else "rbf"
g = torch.Generator().manual_seed(0)
mk = lambda: (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = mk(), mk()
k = sigkernel_amd.RBFKernel(1.0) if kind == "rbf" else sigkernel_amd.LinearKernel()
sk = sigkernel_amd.SigKernel(k, d)
def step():
    Xg = X.clone().requires_grad_(True)
    t0 = time.perf_counter(); mmd = sk.compute_mmd(Xg, Y); torch.cuda.synchronize(); t1 = time.perf_counter()
    mmd.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return t1 - t0, t2 - t1, float(mmd.detach())
step()
f, b, v = step()
pairs = 3 * A * A
print("mmd %s A=%d len=%d dim=%d d=%d: fwd %.1f ms (3 Grams, %.2e pair-solves/s)  bwd %.1f ms  value %.6f  peak mem %.1f GB"
      % (kind, A, M, D, d, f * 1e3, pairs / f, b * 1e3, v, torch.cuda.max_memory_allocated() / 1e9))
