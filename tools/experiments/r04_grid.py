#!/usr/bin/env python3
"""A coarse grid over the public API's inputs (static kernel x path dim x length x dyadic order x dtype): ms per compute_Gram and per
compute_Gram + backward, with the route taken -- a look for performance cliffs outside the BASELINE shapes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
g = torch.Generator().manual_seed(0)
be = _lib.get_backend()
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def t(f, n=3):
    f(); f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
R = {0: "S", 1: "F", 2: "MB", 3: "MBs"}
print("kind   dim len  d dtype |  forward ms (cells/s) route | fwd+bwd ms  route")
for kern in ("linear", "rbf"):
    for D in (1, 3, 6, 12, 20, 40):
        for M in (24, 100, 300):
            for d in (0, 1, 2, 3):
                if d == 3 and M > 100: continue
                for dt in (torch.float64, torch.float32):
                    A = 256 if M <= 100 else 96
                    X, Y = walk(A, M, D).to(dt), walk(A, M, D).to(dt)
                    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0) if kern == "rbf" else sigkernel_amd.LinearKernel(), d)
                    w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda().to(dt)
                    Xg = X.clone().requires_grad_(True)
                    def fb():
                        Xg.grad = None; (sk.compute_Gram(Xg, Y) * w).sum().backward()
                    tf, tb = t(lambda: sk.compute_Gram(X, Y)), t(fb)
                    cells = A * A * ((M - 1) << d) ** 2
                    k = 0 if kern == "linear" else 1
                    print("%-6s %3d %3d %2d %-5s | %9.3f (%.2e) %-3s | %9.3f %-3s  x%.1f" % (kern, D, M, d, "f64" if dt == torch.float64 else "f32", tf, cells / tf * 1e3,
                          R[be.route(_lib.OP_FORWARD, k, D, M, M, d, False, X.element_size())], tb, R[be.route(_lib.OP_ADJOINT, k, D, M, M, d, False, X.element_size())], tb / tf), flush=True)
