#!/usr/bin/env python3
"""Multi-band fused RBF adjoint (sk_rbf_adjoint_fused_mb_f64) against the oracle on small long-path cases."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import _lib
from oracle import oracle as O
be = _lib.get_backend()
def walk(g, A, M, D): return torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)
cases = [(2, 3, 4, 150, 170, 3, "f32"), (2, 2, 3, 140, 200, 16, "f32"), (2, 3, 4, 150, 170, 3), (2, 2, 3, 60, 200, 10), (1, 3, 2, 300, 180, 5), (2, 2, 2, 129, 161, 16), (1, 2, 2, 130, 300, 12), (2, 5, 7, 64, 165, 4)]
if len(sys.argv) > 1: cases = cases[:int(sys.argv[1])]
for case in cases:
    d, A, B, M, N, D = case[:6]
    g = torch.Generator().manual_seed(M + N)
    X, Y = walk(g, A, M, D), walk(g, B, N, D)
    if len(case) > 6: X, Y = X.float(), Y.float()
    w = torch.randn(A, B, generator=g, dtype=torch.float64)
    k = sigkernel_amd.RBFKernel(0.7)
    res = be.solve_fwd_fused_static(1, 0.7, X.cuda(), Y.cuda(), d, False, True, keep_edges=True)
    assert res is not None, "forward unsupported"
    K, edges = res
    Kw = O.gram_forward(X.double(), Y.double(), k, d)
    print("case", case, "fwd rel err %.2e" % float(np.abs(K.cpu().numpy() - Kw).max() / np.abs(Kw).max()), "edges", None if edges is None else tuple(edges.shape))
    out = be.rbf_adjoint_fused_mb(X.cuda(), Y.cuda(), 0.7, d, edges, w.reshape(-1).to(X.dtype).cuda(), gram=True)
    assert out is not None, "adjoint unsupported"
    gr, res_ = out
    want = O.gram_grad_weighted(X.double(), Y.double(), w.numpy(), k, d, nthreads=8)
    got = gr.double().cpu().numpy()
    print("   grad rel err %.3e  residual %.2e" % (float(np.abs(got - want).max() / np.abs(want).max()), float(res_)))
    rows = np.abs(got - want).max(axis=(0, 2)) / np.abs(want).max()
    bad = np.nonzero(rows > 1e-9)[0]
    if len(bad): print("   bad node rows:", bad[:20], "...", bad[-5:], "count", len(bad))
