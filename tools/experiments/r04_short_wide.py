#!/usr/bin/env python3
"""Short paths of dim 5..8 (8 staged dims: the one-band fused forward holds 0.5-1 wave per SIMD there, its LDS rings are per pair
and a wave carries 4-8 pairs): fused route against the streaming route (sk_static_increments + sk_solve_fwd), 2048 x 2048 pairs."""
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
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
for kind in (0, 1):
    for D in (4, 8):
        for M in (16, 32, 33, 64, 65):
            for d in (0, 1):
                A = 2048 if M <= 33 else 1024
                X, Y = walk(A, M, D), walk(A, M, D)
                k = sigkernel_amd.LinearKernel() if kind == 0 else sigkernel_amd.RBFKernel(1.0)
                sk = sigkernel_amd.SigKernel(k, d)
                tf, Kf = t(lambda: sk.compute_Gram(X, Y))
                def streamed():
                    K = torch.empty(A, A, dtype=X.dtype, device=X.device)
                    step = max(1, int(8e9 // (A * M * M * 8)))
                    for a0 in range(0, A, step):
                        inc = be.static_increments(kind, 1.0, X[a0:a0 + step].contiguous(), Y, True)
                        K[a0:a0 + step] = be.solve_fwd(inc, d, False)
                    return K
                ts, Ks = t(streamed)
                err = float((Kf - Ks).abs().max() / Ks.abs().max())
                print("%-6s dim %d len %3d d=%d %4d^2 pairs: fused %7.3f ms  streamed %7.3f ms  (%.2fx)  diff %.1e  route %d" % (
                    "linear" if kind == 0 else "rbf", D, M, d, A, tf, ts, tf / ts, err, be.route(_lib.OP_FORWARD, kind, D, M, M, d, False, 8)), flush=True)
