#!/usr/bin/env python3
"""Short paths with a gradient: compute_Gram + weighted backward through the fused adjoints against the streaming adjoint
(routes.no_fused_adjoint), 1024 x 1024 pairs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
be = _lib.get_backend()
A = 1024
w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
def t(f, n=4):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
for kind in (0, 1):
    for D in (2, 4, 8):
        for M in (16, 32, 64, 100):
            for d in (0, 1):
                X, Y = walk(A, M, D), walk(A, M, D)
                k = sigkernel_amd.LinearKernel() if kind == 0 else sigkernel_amd.RBFKernel(1.0)
                sk = sigkernel_amd.SigKernel(k, d)
                def step():
                    Xg = X.clone().requires_grad_(True)
                    (sk.compute_Gram(Xg, Y) * w).sum().backward()
                    return Xg.grad
                sigkernel_amd.routes.no_fused_adjoint = False
                tf, gf = t(step)
                sigkernel_amd.routes.no_fused_adjoint = True
                ts, gs = t(step)
                sigkernel_amd.routes.no_fused_adjoint = False
                err = float((gf - gs).abs().max() / gs.abs().max())
                print("%-6s dim %d len %3d d=%d: default %8.3f ms  streaming adjoint %8.3f ms  (%.2fx)  diff %.1e  adjoint route %d" % (
                    "linear" if kind == 0 else "rbf", D, M, d, tf, ts, tf / ts, err, be.route(_lib.OP_ADJOINT, kind, D, M, M, d, False, 8)), flush=True)
