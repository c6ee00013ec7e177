#!/usr/bin/env python3
"""Where a C1-sized compute_Gram call spends its host time (cProfile, 2000 calls) and what the device does in it."""
import os, sys, time, cProfile, pstats, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
gen = torch.Generator().manual_seed(0)
walk = lambda A, M, D: (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = walk(5, 10, 2), walk(5, 20, 2)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.5), 1)
for _ in range(50): sk.compute_Gram(X, Y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): sk.compute_Gram(X, Y)
torch.cuda.synchronize(); print("%.1f us/call" % ((time.perf_counter() - t0) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): sk.compute_Gram(X, Y)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
