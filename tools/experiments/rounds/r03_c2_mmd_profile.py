#!/usr/bin/env python3
"""Kernel table of a C2-sized training step: compute_mmd(X, Y).backward(), 128 + 128 paths, len 64, dim 3, RBF, d = 1."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
gen = torch.Generator().manual_seed(0)
walk = lambda A, M, D: (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = walk(128, 64, 3), walk(128, 64, 3)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
def step():
    Xg = X.detach().requires_grad_(True)
    sk.compute_mmd(Xg, Y).backward()
    return Xg.grad
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print("%.1f us/step" % ((time.perf_counter() - t0) / 200 * 1e6))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(20): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=90))
