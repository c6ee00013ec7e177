#!/usr/bin/env python3
"""fp32, dyadic 2, long paths WITH gradients (BASELINE configs[4] shape, reduced batch): compute_mmd + backward.
usage: python tools/time_c5grad.py [batch] [len] [dim]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
A = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
D = int(sys.argv[3]) if len(sys.argv) > 3 else 16
g = torch.Generator().manual_seed(0)
mk = lambda: (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).float().cuda()
X, Y = mk(), mk()
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2)
def step():
    Xg = X.clone().requires_grad_(True)
    t0 = time.perf_counter(); mmd = sk.compute_mmd(Xg, Y); torch.cuda.synchronize(); t1 = time.perf_counter()
    mmd.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return t1 - t0, t2 - t1, Xg.grad
step()
f, b, gr = step()
# reference gradient: the same computation in fp64
sk64 = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2)
Xd = X.double().requires_grad_(True)
sk64.compute_mmd(Xd, Y.double()).backward()
err = float((gr.double() - Xd.grad).abs().max() / Xd.grad.abs().max())
print("mmd fp32 d=2 A=%d len=%d dim=%d: fwd %.1f ms  bwd %.1f ms  grad rel err vs fp64 run %.1e" % (A, M, D, f * 1e3, b * 1e3, err))
