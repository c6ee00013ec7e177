#!/usr/bin/env python3
"""compute_Gram on SHORT paths (10..64 points: what most users of the reference feed it) -- cells/s against the 4e12 of the headline."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
for kern in ("linear", "rbf"):
    for (A, M, N, D, d) in ((2048, 10, 20, 2, 1), (2048, 16, 16, 2, 0), (2048, 16, 16, 2, 1), (2048, 32, 32, 4, 0), (2048, 32, 32, 4, 1), (2048, 33, 33, 4, 1), (2048, 64, 64, 4, 1), (2048, 64, 64, 4, 0), (1024, 128, 128, 4, 1)):
        X, Y = walk(A, M, D), walk(A, N, D)
        sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0) if kern == "rbf" else sigkernel_amd.LinearKernel(), d)
        for _ in range(3): K = sk.compute_Gram(X, Y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): K = sk.compute_Gram(X, Y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        cells = A * A * ((M - 1) << d) * ((N - 1) << d)
        print("%-6s %4d x %4d pairs, len %3d x %3d, dim %d, d=%d: %8.3f ms  %.2e pairs/s  %.2e cells/s" % (kern, A, A, M, N, D, d, dt * 1e3, A * A / dt, cells / dt), flush=True)
