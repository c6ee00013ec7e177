#!/usr/bin/env python3
"""The reference's example workload (examples/time_series_classification.py:186-197): RBF, dyadic_order 0, lead-lag + time paths
(dim 5..8) of a few hundred points; compute_Gram(X, X, sym=True) for the training set and compute_Gram(X_test, X_train)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
be = _lib.get_backend()
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (A, M, D, d) in ((300, 199, 5, 0), (300, 297, 7, 0), (1000, 297, 7, 0), (300, 599, 7, 0), (300, 100, 5, 0), (300, 60, 7, 0), (300, 297, 7, 1)):
    X, Xt = walk(A, M, D), walk(A // 3, M, D)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), d)
    cells = ((M - 1) << d) ** 2
    ts = t(lambda: sk.compute_Gram(X, X, sym=True))
    tf = t(lambda: sk.compute_Gram(X, X))
    tt = t(lambda: sk.compute_Gram(Xt, X))
    print("A=%4d len %3d dim %d d=%d route %d: sym %8.3f ms (%.2e cells/s on the triangle), all pairs %8.3f ms (%.2e cells/s), test x train %8.3f ms" % (
        A, M, D, d, be.route(_lib.OP_FORWARD, 1, D, M, M, d, False, 8), ts, A * (A + 1) / 2 * cells / ts * 1e3, tf, A * A * cells / tf * 1e3, tt), flush=True)
