#!/usr/bin/env python3
"""Long first paths against short second ones with a gradient, LinearKernel (and RBF beside it): the swapped one-band adjoint with
second-argument sums (route FUSED_SWAP, round 6 for the linear kernel) against the routes without the swap (routes.no_adjoint_swap).
usage: r06_asym.py  -> profiles/r06_asym.txt"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
be = _lib.get_backend()
R = {0: "S", 1: "F", 2: "MB", 3: "MBs", 4: "Fs"}
def t(f, n=5, reps=5):     # median of `reps` timings of n steps each (one-off allocator / first-launch costs fall out)
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2], r
kinds = sys.argv[1:] or ("linear", "rbf")
RQ = sigkernel_amd.sigkernel._route_query
for kind in kinds:
    for A in (128, 32):
        w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
        for D in ((3, 8) if kind == "linear" else (3,)):
            for d in (0, 1, 2):
                for M, N in ((512, 64), (300, 40), (700, 80), (800, 100), (1500, 90), (1000, 100), (2000, 128), (200, 9), (66, 30), (140, 64), (130, 128)):
                    kk = 0 if kind == "linear" else 1
                    # (shapes the route table would leave to the multi-band adjoint are run with the swapped route forced, marked *: none
                    # since the carry moved from LDS to DPP -- the first build kept LinearKernel at dyadic 0 with efficient bands there)
                    forced = be.route(_lib.OP_ADJOINT, kk, D, M, N, d, False, 8) != _lib.ROUTE_FUSED_SWAP
                    if forced and not (kind == "linear" and d == 0 and N <= 129 and be.route(_lib.OP_ADJOINT, kk, D, M, N, d, False, 8, True) == _lib.ROUTE_FUSED_MB): continue
                    X, Y = walk(A, M, D), walk(A, N, D)
                    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
                    sk = sigkernel_amd.SigKernel(k, d)
                    def step():
                        Xg = X.clone().requires_grad_(True)
                        (sk.compute_Gram(Xg, Y) * w).sum().backward()
                        return Xg.grad
                    res = []
                    for off in (False, True):
                        sigkernel_amd.routes.no_adjoint_swap = off
                        RQ.cache_clear()
                        if forced and not off:
                            sigkernel_amd.sigkernel._route_query = lambda fn, op, *key: _lib.ROUTE_FUSED_SWAP if op == _lib.OP_ADJOINT else RQ(fn, op, *key)
                        else:
                            sigkernel_amd.sigkernel._route_query = RQ
                        tg, gr = t(step)
                        res.append((tg, gr))
                    sigkernel_amd.routes.no_adjoint_swap = False
                    sigkernel_amd.sigkernel._route_query = RQ
                    RQ.cache_clear()
                    err = float((res[0][1] - res[1][1]).abs().max() / res[1][1].abs().max())
                    print("%-6s %3d x %-3d paths dim %d d=%d %4d x %-4d points%s | fwd+bwd swapped %7.3f ms  without the swap %7.3f ms  ratio %.2f | grad diff %.1e"
                          % (kind, A, A, D, d, M, N, "*" if forced else " ", res[0][0], res[1][0], res[0][0] / res[1][0], err), flush=True)
