#!/usr/bin/env python3
"""Where a README-sized compute_mmd(X, Y).backward() spends its host time (cProfile, 500 steps) and which kernels it launches."""
import os, sys, time, cProfile, pstats, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
gen = torch.Generator().manual_seed(0)
walk = lambda A, M, D: (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = walk(5, 10, 2), walk(5, 20, 2)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.5), 1)
def step():
    Xg = X.detach().requires_grad_(True)
    sk.compute_mmd(Xg, Y).backward()
    return Xg.grad
for _ in range(30): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500): step()
torch.cuda.synchronize(); print("%.1f us/step" % ((time.perf_counter() - t0) / 500 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32); print(s.getvalue()[:6000])
if len(sys.argv) > 1:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(20): step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70)[:9000])
