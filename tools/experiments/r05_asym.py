#!/usr/bin/env python3
"""Long first paths against short second ones (and the other way round), with a gradient: the default route against the fused routes
forced (routes.no_stream) -- where does swapping X and Y change the time?  usage: r05_asym.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
be = _lib.get_backend()
R = {0: "S", 1: "F", 2: "MB", 3: "MBs", 4: "Fs"}
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
A = 128
w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
for kind in ("linear", "rbf"):
    for D in (3, 8, 12):
        for d in (0, 1):
            for M, N in ((512, 64), (64, 512), (300, 40), (40, 300), (1000, 100), (100, 1000)):
                X, Y = walk(A, M, D), walk(A, N, D)
                k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
                sk = sigkernel_amd.SigKernel(k, d)
                def step():
                    Xg = X.clone().requires_grad_(True)
                    (sk.compute_Gram(Xg, Y) * w).sum().backward()
                    return Xg.grad
                def fwd(): return sk.compute_Gram(X, Y)
                res = []
                for ns in (False, True):
                    sigkernel_amd.routes.no_stream = ns
                    sigkernel_amd.sigkernel._route_query.cache_clear()
                    tf, _ = t(fwd)
                    tg, gr = t(step)
                    kk = 0 if kind == "linear" else 1
                    res.append((tf, tg, gr, R[be.route(_lib.OP_FORWARD, kk, D, M, N, d, False, 8, ns)], R[be.route(_lib.OP_ADJOINT, kk, D, M, N, d, False, 8, ns)]))
                sigkernel_amd.routes.no_stream = False
                err = float((res[0][2] - res[1][2]).abs().max() / res[0][2].abs().max())
                print("%-6s dim %2d d=%d %4d x %-4d | default fwd %7.2f (%3s) fwd+bwd %8.2f (%3s) | fused forced fwd %7.2f (%3s) fwd+bwd %8.2f (%3s) | grad diff %.1e"
                      % (kind, D, d, M, N, res[0][0], res[0][3], res[0][1], res[0][4], res[1][0], res[1][3], res[1][1], res[1][4], err), flush=True)
