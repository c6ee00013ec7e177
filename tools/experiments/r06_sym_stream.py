#!/usr/bin/env python3
"""compute_Gram(X, X, sym=True) WITHOUT a gradient on the streaming route (wide paths: the example pipeline's SVC Gram matrices on lead-lag
paths): one block (all pairs) against the blocked triangle, by batch size -- where does the triangle start to pay when the static
kernel, not the solver, is most of a pair's cost?  usage: r06_sym_stream.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import sigkernel as S
g = torch.Generator().manual_seed(0)
def walk(A, M, D, dt=torch.float64): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()
def t(f, n=4, reps=5):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2], r
for kind, D, M, d in (("rbf", 20, 64, 1), ("linear", 20, 64, 1), ("rbf", 12, 64, 1), ("rbf", 20, 128, 0), ("rbf", 3, 64, 3)):
    k = sigkernel_amd.RBFKernel(1.0) if kind == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, d)
    for A in (64, 128, 192, 256, 384, 512):
        X = walk(A, M, D)
        res = []
        for cells in (1e30, 0.0):
            S._SYM_MIN_CELLS = cells
            res.append(t(lambda: sk.compute_Gram(X, X, sym=True)))
        S._SYM_MIN_CELLS = None
        err = float((res[0][1] - res[1][1]).abs().max())
        cells = float(A) * A * ((M - 1) << d) ** 2
        print("%-6s dim %2d %3d points d=%d %4d paths (%.1e cells) | one block %7.3f ms | blocked triangle %7.3f ms | ratio %.2f | diff %.1e"
              % (kind, D, M, d, A, cells, res[0][0], res[1][0], res[1][0] / res[0][0], err), flush=True)
