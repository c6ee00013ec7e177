#!/usr/bin/env python3
"""Wide determinism sweep: static kernel x dyadic order x path dim x lengths x precision x API call, four runs each, bitwise
comparison (a variant whose reads race with register spills shows up as run-to-run differences or NaN)."""
import itertools, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
def walk(g, A, M, D, dt): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt)
bad = 0; n = 0
for kname, d, D, (M, N), dt, naive in itertools.product(("linear", "rbf"), (0, 1, 2), (2, 4, 6, 8, 12), ((20, 33), (64, 64), (33, 170), (130, 129), (257, 161)),
                                                       (torch.float64, torch.float32), (False, True)):
    if naive and (M > 64 or D > 8): continue
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
        out = [K.detach(), Xg.grad, m.detach().reshape(1), Xs.grad]
        if not naive and M <= 130: out += list(sk.compute_kernel_and_derivatives_Gram(X, Y, gam))
        return torch.cat([t.double().flatten() for t in out])
    runs = [once() for _ in range(4)]
    n += 1
    nan = bool(torch.isnan(runs[0]).any())
    same = all(torch.equal(r, runs[0]) for r in runs[1:])
    if nan or not same:
        bad += 1
        print("NONDETERMINISTIC" if not same else "NAN", kname, "d", d, "D", D, (M, N), str(dt)[6:], "naive" if naive else "", flush=True)
print("%d combinations, %d bad" % (n, bad))
