#!/usr/bin/env python3
"""compute_Gram(X, X, sym=True) with a gradient, LinearKernel: the triangle through the one-band adjoint with BOTH sets of sums (round 6)
against all pairs (sk_route_query(SK_OP_ADJOINT_SYM) answered STREAM: routes.no_sym_fused).  usage: r06_sym_linear.py -> profiles/r06_sym_linear.txt"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib, sigkernel as S
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def t(f, n=3, reps=5):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2], r
RQ = S._route_query
SHAPES = ((512, 128, 8, 1), (1024, 64, 4, 1), (512, 64, 3, 2), (1024, 128, 6, 0), (2048, 32, 3, 1))
if len(sys.argv) > 1 and sys.argv[1] == "rbf-batch":     # where does the RBF triangle start to pay?  (the gate: cost entry sym_min_cells)
    SHAPES = tuple((A, M, 3, d) for M, d in ((64, 2), (64, 1), (128, 0), (32, 1)) for A in (384, 512, 640, 768, 1024, 1536))
for kind in (("rbf",) if len(sys.argv) > 1 else ("linear", "rbf")):
    for A, M, D, d in SHAPES:
        if kind == "rbf" and (D > 4 or (d >= 1 and M > 64)): continue
        X = walk(A, M, D)
        w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
        sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0), d)
        def step():
            Xg = X.clone().requires_grad_(True)
            (sk.compute_Gram(Xg, Xg, sym=True) * w).sum().backward()
            return Xg.grad
        res = []
        for off in (False, True):
            S._route_query = (lambda fn, op, *key: _lib.ROUTE_STREAM if op == _lib.OP_ADJOINT_SYM else RQ(fn, op, *key)) if off else RQ
            res.append(t(step))
        S._route_query = RQ
        err = float((res[0][1] - res[1][1]).abs().max() / res[1][1].abs().max())
        print("%-6s %4d paths of %3d points dim %d d=%d | sym Gram + backward: triangle with both sums %8.3f ms  all pairs %8.3f ms  ratio %.2f | grad diff %.1e"
              % (kind, A, M, D, d, res[0][0], res[1][0], res[0][0] / res[1][0], err), flush=True)
