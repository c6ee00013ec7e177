#!/usr/bin/env python3
"""Per-pair cost of Gram calls at extreme batch aspect ratios, and of big paired batches, against the square case (64 points, d=1).
usage: r06_aspect.py -> profiles/r06_aspect.txt"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def t(f, n=3, reps=3):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2]
for kind, D in (("linear", 8), ("rbf", 3)):
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
    sk = sigkernel_amd.SigKernel(k, 1)
    for A, B in ((512, 512), (1, 262144), (262144, 1), (4, 65536), (65536, 4), (16, 16384), (16384, 16), (64, 4096), (4096, 64), (3, 50001), (50001, 3)):
        X, Y = walk(A, 64, D), walk(B, 64, D)
        w = torch.randn(A, B, generator=g, dtype=torch.float64).cuda()
        def fwd(): sk.compute_Gram(X, Y)
        def step():
            Xg = X.clone().requires_grad_(True); (sk.compute_Gram(Xg, Y) * w).sum().backward()
        tf, ts = t(fwd), t(step)
        print("%-6s Gram %6d x %-6d | forward %8.3f ms %6.2f ns/pair | forward + backward %8.3f ms %6.2f ns/pair" % (kind, A, B, tf, tf * 1e6 / (A * B), ts, ts * 1e6 / (A * B)), flush=True)
    for A in (4096, 262144):
        X, Y = walk(A, 64, D), walk(A, 64, D)
        def fwd(): sk.compute_kernel(X, Y)
        def step():
            Xg = X.clone().requires_grad_(True); sk.compute_kernel(Xg, Y).sum().backward()
        tf, ts = t(fwd), t(step)
        print("%-6s paired %6d          | forward %8.3f ms %6.2f ns/pair | forward + backward %8.3f ms %6.2f ns/pair" % (kind, A, tf, tf * 1e6 / A, ts, ts * 1e6 / A), flush=True)
