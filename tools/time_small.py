#!/usr/bin/env python3
"""Latency of small calls (BASELINE configs[0] and [1]): wall time per call vs GPU kernel time.
usage: python tools/time_small.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
gen = torch.Generator().manual_seed(0)
walk = lambda A, M, D: (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def bench(label, f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-58s %8.1f us/call" % (label, dt * 1e6))
X, Y = walk(5, 10, 2), walk(5, 20, 2)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.5), 1)
bench("c1 compute_Gram 5x5 len 10/20 rbf d=1", lambda: sk.compute_Gram(X, Y))
bench("c1 compute_kernel 5 len 10/20", lambda: sk.compute_kernel(X, Y))
def fb():
    Xg = X.clone().requires_grad_(True); sk.compute_mmd(Xg, Y).backward()
bench("c1 compute_mmd + backward", fb, 50)
X2 = walk(128, 64, 3)
sk2 = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
bench("c2 compute_Gram 128x128 len 64 rbf d=1 sym", lambda: sk2.compute_Gram(X2, X2, sym=True))
bench("c2 compute_Gram 128x128 len 64 rbf d=1", lambda: sk2.compute_Gram(X2, X2))
sk3 = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1)
bench("c2-shape linear (fused kernel)", lambda: sk3.compute_Gram(X2, X2))
def fb2():
    Xg = X2.clone().requires_grad_(True); sk2.compute_mmd(Xg, X2).backward()
bench("c2 compute_mmd + backward", fb2, 20)
