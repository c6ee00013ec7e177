#!/usr/bin/env python3
"""Determinism of every route, bit for bit (a variant whose in-flight reads race with register spills or copies shows up as
run-to-run differences or NaN: commit 6b76349).

    python tools/det_sweep.py              wide: static kernel x dyadic order x path dim x lengths x precision x stencil, 4 runs each
    python tools/det_sweep.py --families   deep: one shape per inline-asm kernel family (one-band / multi-band forward and adjoints of
                                           both static kernels incl. rbf at dyadic 0 and the fp32 ring, the triangular second-argument
                                           adjoint, streaming solver + adjoint, derivative solvers), 100 runs each
"""
import itertools, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
def walk(g, A, M, D, dt): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt)
FAMILIES = "--families" in sys.argv
if FAMILIES:
    RUNS = 100
    combos = [("linear", 1, 8, (64, 64), torch.float64, False), ("linear", 0, 4, (129, 40), torch.float64, True),
              ("rbf", 2, 4, (64, 64), torch.float64, False), ("rbf", 1, 3, (40, 41), torch.float32, True),
              ("rbf", 0, 4, (100, 90), torch.float64, False), ("rbf", 0, 6, (130, 129), torch.float64, False),
              ("rbf", 0, 12, (60, 170), torch.float64, True), ("linear", 1, 12, (257, 161), torch.float64, False),
              ("linear", 2, 6, (150, 20), torch.float32, False), ("rbf", 1, 12, (130, 129), torch.float64, False),
              ("rbf", 2, 12, (150, 161), torch.float32, False), ("rbf", 1, 7, (64, 64), torch.float64, False),
              ("linear", 1, 20, (40, 33), torch.float64, False), ("rbf", 2, 20, (33, 40), torch.float32, False),
              # few pairs of long paths: the bands of a pair on several waves (sk_wave_fused_mb.hip, split mode: the call without a gradient)
              ("linear", 1, 6, (300, 520), torch.float64, False), ("rbf", 0, 3, (600, 530), torch.float64, False),
              ("rbf", 2, 12, (150, 512), torch.float32, False)]
else:
    RUNS = 4
    combos = itertools.product(("linear", "rbf"), (0, 1, 2), (2, 4, 6, 8, 12), ((20, 33), (64, 64), (33, 170), (130, 129), (257, 161)),
                               (torch.float64, torch.float32), (False, True))
bad = 0; n = 0
for kname, d, D, (M, N), dt, naive in combos:
    g = torch.Generator().manual_seed(M * 3 + N + D)
    A, B = 5, 6
    X, Y = walk(g, A, M, D, dt).cuda(), walk(g, B, N, D, dt).cuda()
    gam = torch.randn(A, M, D, generator=g).to(dt).cuda()
    w = torch.randn(A, B, generator=g).to(dt).cuda()
    k = sigkernel_amd.RBFKernel(0.9) if kname == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, d, _naive_solver=naive)
    def once():
        Xg = X.clone().requires_grad_(True)
        K = sk.compute_Gram(Xg, Y); (K * w).sum().backward()
        Xs = X.clone().requires_grad_(True)
        m = sk.compute_mmd(Xs, Y); m.backward()
        out = [K.detach(), Xg.grad, m.detach().reshape(1), Xs.grad, sk.compute_Gram(X, Y)]
        if not naive and M <= 130: out += list(sk.compute_kernel_and_derivatives_Gram(X, Y, gam))
        return torch.cat([t.double().flatten() for t in out])
    first = once()
    n += 1
    nan = bool(torch.isnan(first).any())
    same = all(torch.equal(once(), first) for _ in range(RUNS - 1))
    if nan or not same:
        bad += 1
        print("NONDETERMINISTIC" if not same else "NAN", kname, "d", d, "D", D, (M, N), str(dt)[6:], "naive" if naive else "", flush=True)
print("%d combinations x %d runs, %d bad" % (n, RUNS, bad))
sys.exit(1 if bad else 0)
