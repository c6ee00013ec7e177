#!/usr/bin/env python3
"""Does swapping X and Y change the time of a Gram with a gradient?  fwd+bwd of compute_Gram(X, Y) (gradient with respect to X) with
long first / short second paths against the other way round, default routes; the reference's cost is symmetric in its two arguments
(sigkernel.py:419-502).  VERDICT r5 item 5: no cell above 1.5x.  usage: r06_asym_xy.py  -> profiles/r06_asym_xy.txt"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
be = _lib.get_backend()
R = {0: "S", 1: "F", 2: "MB", 3: "MBs", 4: "Fs"}
def t(f, n=5, reps=5):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2]
A = 128
w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
worst = 0.0
for kind in ("linear", "rbf"):
    for D in (3, 6, 8, 12):
        for d in (0, 1, 2):
            for M, N in ((512, 64), (300, 40), (1000, 100), (200, 20)):
                if d == 2 and M > 512: continue
                k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
                sk = sigkernel_amd.SigKernel(k, d)
                res = []
                for m, n in ((M, N), (N, M)):
                    X, Y = walk(A, m, D), walk(A, n, D)
                    def step():
                        Xg = X.clone().requires_grad_(True)
                        (sk.compute_Gram(Xg, Y) * w).sum().backward()
                    kk = 0 if kind == "linear" else 1
                    res.append((t(step), R[be.route(_lib.OP_ADJOINT, kk, D, m, n, d, False, 8)]))
                r = res[0][0] / res[1][0]
                worst = max(worst, r, 1 / r)
                print("%-6s dim %2d d=%d | %4d x %-4d points %8.3f ms (%3s) | %4d x %-4d points %8.3f ms (%3s) | ratio %.2f%s"
                      % (kind, D, d, M, N, res[0][0], res[0][1], N, M, res[1][0], res[1][1], r, "   > 1.5" if max(r, 1 / r) > 1.5 else ""), flush=True)
print("worst ratio %.2f" % worst)
