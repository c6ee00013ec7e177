#!/usr/bin/env python3
"""Multi-band fused LINEAR adjoint (sk_linear_adjoint_fused_mb_f64) against the oracle on small long-path cases, and timings."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import _lib
from oracle import oracle as O
be = _lib.get_backend()
def walk(g, A, M, D): return torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)
cases = [(1, 3, 4, 300, 170, 3), (0, 2, 3, 300, 200, 10), (2, 3, 2, 150, 180, 5), (1, 2, 2, 129, 161, 16), (0, 2, 2, 257, 300, 12), (2, 5, 7, 64, 165, 4), (1, 2, 3, 140, 161, 8)]
for d, A, B, M, N, D in cases:
    g = torch.Generator().manual_seed(M + N)
    X, Y = walk(g, A, M, D), walk(g, B, N, D)
    w = torch.randn(A, B, generator=g, dtype=torch.float64)
    k = sigkernel_amd.LinearKernel()
    res = be.solve_fwd_fused_static(0, 1.0, X.cuda(), Y.cuda(), d, False, True, keep_edges=True)
    assert res is not None and res[1] is not None, "forward unsupported"
    K, edges = res
    Kw = O.gram_forward(X, Y, k, d)
    print("case", (d, A, B, M, N, D), "fwd rel err %.2e" % float(np.abs(K.cpu().numpy() - Kw).max() / np.abs(Kw).max()), flush=True)
    out = be.linear_adjoint_fused_mb(X.cuda(), Y.cuda(), 1.0, d, edges, w.reshape(-1).cuda(), gram=True)
    assert out is not None, "adjoint unsupported"
    want = O.gram_grad_weighted(X, Y, w.numpy(), k, d, nthreads=8)
    got = out[0].cpu().numpy()
    print("   grad rel err %.3e  residual %.2e" % (float(np.abs(got - want).max() / np.abs(want).max()), float(out[1])), flush=True)
if len(sys.argv) > 1:
    for (A, M, D, d) in ((256, 512, 8, 1), (256, 300, 8, 0), (512, 200, 4, 2)):
        g = torch.Generator().manual_seed(0)
        X, Y = walk(g, A, M, D).cuda(), walk(g, A, M, D).cuda()
        w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
        sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), d)
        def step():
            Xg = X.clone().requires_grad_(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            K = sk.compute_Gram(Xg, Y); torch.cuda.synchronize(); t1 = time.perf_counter()
            (K * w).sum().backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
            return t1 - t0, t2 - t1, Xg.grad
        outs = {}
        for tag, env in (("fused multi-band", None), ("unfused", "SK_NO_FUSED_ADJOINT")):
            if env: os.environ[env] = "1"; sigkernel_amd.routes.reload()
            step()
            best = min((step() for _ in range(3)), key=lambda r: r[0] + r[1])
            outs[tag] = best[2]
            print("Linear %dx%d len %d dim %d d=%d %-17s fwd %.1f ms  bwd %.1f ms" % (A, A, M, D, d, tag, best[0] * 1e3, best[1] * 1e3), flush=True)
            if env: del os.environ[env]; sigkernel_amd.routes.reload()
        a, b = outs["fused multi-band"], outs["unfused"]
        print("   gradients: max rel diff %.2e" % float((a - b).abs().max() / b.abs().max()))
