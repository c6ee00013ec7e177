#!/usr/bin/env python3
"""Multi-band fused kernels against the streaming route around the efficiency thresholds of the cost table (128 x 128 pairs)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib, sigkernel as S
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def ms(f, n=5):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
R = sigkernel_amd.routes
def eff(kind, M, N, d, rc=None):
    rcs = {0: 4, 1: 2, 2: 1}; rc = rc or rcs[d]
    rows = M - 1 + (1 if kind == "rbf" else 0); nb = -(-rows // (64 * rc)); nu = (N + 1) // 2 if kind == "rbf" else N // 2
    return rows / (nb * 64 * rc) * nu / max(80, -(-nu // 8) * 8)
A = 128
w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
CASES = (("linear", 12, 1), ("linear", 12, 0), ("rbf", 12, 1), ("rbf", 7, 1), ("rbf", 7, 0), ("linear", 12, 2))
SIZES = (80, 100, 110, 120, 128, 140, 160, 200, 256)
if len(sys.argv) > 1:      # second pass: 16 staged dims at dyadic 2 / 0, longer paths
    CASES, SIZES = (("rbf", 16, 2), ("rbf", 16, 0), ("rbf", 12, 2), ("linear", 12, 2), ("linear", 16, 0)), (129, 140, 200, 300, 512)
    A = 64
    w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
for kind, D, d in CASES:
    for M in SIZES:
        X, Y = walk(A, M, D), walk(A, M, D)
        sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0), d)
        def fwd(): sk.compute_Gram(X, Y)
        def grad():
            Xg = X.clone().requires_grad_(True); (sk.compute_Gram(Xg, Y) * w).sum().backward()
        res = []
        for mb in (True, False):
            R.no_stream, R.no_fused_mb = mb, not mb
            S._route_query.cache_clear()
            res.append((ms(fwd), ms(grad)))
        R.no_stream = R.no_fused_mb = False
        S._route_query.cache_clear()
        be = _lib.get_backend(); k = 0 if kind == "linear" else 1
        print("%-6s dim %2d d=%d %3d points eff %.2f | forward: multi-band %7.2f streamed %7.2f (%.2f) default %d | with gradient: multi-band %7.2f streamed %7.2f (%.2f) default %d"
              % (kind, D, d, M, eff(kind, M, M, d, 2 if (kind == "rbf" and d == 0) else None), res[0][0], res[1][0], res[0][0] / res[1][0], be.route(0, k, D, M, M, d, False, 8),
                 res[0][1], res[1][1], res[0][1] / res[1][1], be.route(1, k, D, M, M, d, False, 8)), flush=True)
