#!/usr/bin/env python3
"""compute_Gram(X, Y) + backward through the one-band fused adjoints by batch size: a lane group sweeps one CHUNK of the B pairs of an x_a;
until round 6 the chunk length had to DIVIDE B, so batches without a suitable divisor (primes, 640, 768, 1536 ...) left lane groups idle.
usage: r06_batch_divisors.py -> profiles/r06_batch_divisors.txt"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def t(f, n=4, reps=5):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2]
be = _lib.get_backend()
for kind, M, D, d in (("linear", 64, 4, 1), ("rbf", 64, 3, 1), ("linear", 128, 8, 1)):
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
    sk = sigkernel_amd.SigKernel(k, d)
    for A in (96, 101, 127, 128, 131, 250, 251, 256, 509, 512, 600, 640, 768, 1000, 1024, 1280, 1536):
        if M == 128 and A > 600: continue
        X, Y = walk(A, M, D), walk(A, M, D)
        w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
        def fwd(): return sk.compute_Gram(X, Y)
        def step():
            Xg = X.clone().requires_grad_(True)
            (sk.compute_Gram(Xg, Y) * w).sum().backward()
        tf, ts = t(fwd), t(step)
        print("%-6s len %3d dim %d d=%d  %4d x %-4d pairs | forward %8.3f ms (%6.2f ns/pair) | forward + backward %8.3f ms (%6.2f ns/pair) | chunk %s"
              % (kind, M, D, d, A, A, tf, tf * 1e6 / (A * A), ts, ts * 1e6 / (A * A), getattr(be, "last_fused_ppg", None)), flush=True)
