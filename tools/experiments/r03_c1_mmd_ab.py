#!/usr/bin/env python3
"""Wall-clock A/B of a README-sized compute_mmd(X, Y).backward() between two builds (host-bound: ~50 launches per step).
usage: r03_c1_mmd_ab.py <build dir under build_ab | new>   (run alternately in one gpurun call)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, "build_ab", which) if which != "new" else ROOT)
import numpy as np, torch
import sigkernel_amd
gen = torch.Generator().manual_seed(0)
walk = lambda A, M, D: (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = walk(5, 10, 2), walk(5, 20, 2)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.5), 1)
def step():
    Xg = X.detach().requires_grad_(True)
    sk.compute_mmd(Xg, Y).backward()
    return Xg.grad
for _ in range(100): g = step()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): g = step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 300 * 1e6)
print("%-5s c1 mmd+backward: median %.1f us/step  min %.1f   grad checksum %r" % (which, float(np.median(ts)), min(ts), float(g.sum())), flush=True)
