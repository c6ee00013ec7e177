#!/usr/bin/env python3
"""The fused derivative solver (sk_solve_deriv_static_f64) against the unfused route (sk_static_deriv_increments + sk_solve_deriv) on the same
inputs, and timings at the shape of profiles/r01_deriv_kernel.txt (256 x 256 pairs, len 128, dim 8, d = 1)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
def walk(g, A, M, D): return torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)
def run(kern, d, X, Y, gm, fused):
    if fused: os.environ.pop("SK_NO_FUSED_DERIV", None)
    else: os.environ["SK_NO_FUSED_DERIV"] = "1"
    sigkernel_amd.routes.reload()
    return sigkernel_amd.SigKernel(kern, d).compute_kernel_and_derivatives_Gram(X, Y, gm)
cases = [(1, 3, 2, 130, 128, 5), (0, 2, 3, 100, 140, 8), (2, 2, 2, 70, 127, 3), (1, 2, 3, 128, 158, 16), (1, 4, 3, 50, 127, 4), (0, 3, 2, 64, 126, 7), (1, 3, 4, 40, 170, 3), (0, 2, 3, 70, 200, 8), (1, 3, 2, 130, 180, 5), (2, 2, 2, 20, 161, 2), (1, 2, 2, 65, 300, 12), (0, 5, 7, 129, 165, 4), (2, 2, 3, 70, 170, 9)]
for kern in (sigkernel_amd.LinearKernel(), sigkernel_amd.RBFKernel(0.8)):
    for d, A, B, M, N, D in cases:
        g = torch.Generator().manual_seed(M + N)
        X, Y, gm = walk(g, A, M, D).cuda(), walk(g, B, N, D).cuda(), torch.randn(A, M, D, generator=g, dtype=torch.float64).cuda()
        a = run(kern, d, X, Y, gm, True)
        b = run(kern, d, X, Y, gm, False)
        errs = [float((x - y).abs().max() / y.abs().max()) for x, y in zip(a, b)]
        print(type(kern).__name__, (d, A, B, M, N, D), "rel diff fused vs unfused: k %.2e  kd %.2e  kdd %.2e" % tuple(errs), flush=True)
if len(sys.argv) > 1:
    for kern in (sigkernel_amd.LinearKernel(), sigkernel_amd.RBFKernel(1.0)):
        for (A, M, D, d) in ((256, 128, 8, 1), (256, 128, 8, 0), (256, 64, 8, 2), (256, 64, 8, 1), (128, 256, 4, 1)):
            if M < 160 and False: continue
            g = torch.Generator().manual_seed(1)
            X, Y, gm = walk(g, A, M, D).cuda(), walk(g, A, M, D).cuda(), torch.randn(A, M, D, generator=g, dtype=torch.float64).cuda()
            for fused in (True, False):
                for _ in range(2): run(kern, d, X, Y, gm, fused)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): run(kern, d, X, Y, gm, fused)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
                print("%-12s %dx%d len %d/%d dim %d d=%d %s: %.2f ms" % (type(kern).__name__, A, A, M, M, D, d, "fused  " if fused else "unfused", dt * 1e3), flush=True)
