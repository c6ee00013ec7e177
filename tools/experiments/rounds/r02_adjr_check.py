#!/usr/bin/env python3
"""GPU box: the fused RBF adjoint (sk_rbf_adjoint_fused_f64) against the unfused route on random shapes, then timing at C4 / C3-RBF tiles."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import _lib
be = _lib.get_backend(); dev = "cuda:0"
def walk(gen, A, M, D, dtype=torch.float64):
    return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)).to(dtype)
rng = np.random.default_rng(1)
bad = n = 0
for it in range(int(os.environ.get("N_IT", "60"))):
    d = int(rng.integers(1, 3))
    cap = 64 * (4 >> d)
    M = int(rng.integers(2, cap + 1)) if it % 4 else cap
    N = int(rng.integers(2, 150))
    A, B, D = int(rng.integers(1, 20)), int(rng.integers(1, 30)), int(rng.integers(1, 9))
    gram = bool(it % 3)
    if not gram: B = A
    sig = float(rng.uniform(0.5, 1.5))
    gen = torch.Generator().manual_seed(300 + it)
    X, Y = (walk(gen, A, M, D) * 2).to(dev), (walk(gen, B, N, D) * 2).to(dev)
    go = torch.randn(A * B if gram else A, generator=gen, dtype=torch.float64).to(dev) if it % 5 else None
    res = be.solve_fwd_fused_rbf(X, Y, sig, d, False, gram, keep_edges=True)
    if res is None or res[1] is None:
        print("it", it, "no edges", d, M, N, D); continue
    K, edges = res
    got = be.rbf_adjoint_fused(X, Y, sig, d, edges, go, gram=gram)
    if got is None:
        print("it %d unsupported d=%d M=%d N=%d D=%d" % (it, d, M, N, D)); continue
    inc = be.static_increments(1, sig, X, Y, gram)
    _, W = be.solve_adj(inc, d, False, edges=edges)
    want = be.static_adjoint(1, sig, X, Y, W, go, gram)
    err = float((got[0] - want).abs().max() / want.abs().max())
    n += 1
    if not err <= 1e-10:
        bad += 1
        print("MISMATCH it=%d d=%d A=%d B=%d M=%d N=%d D=%d gram=%s go=%s err=%.3e res=%.2e" % (it, d, A, B, M, N, D, gram, go is not None, err, float(got[1])))
print("checked %d, bad %d" % (n, bad))
def tm(f, reps=3):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for (A, B, M, D, d) in ((256, 2048, 64, 4, 2), (256, 2048, 64, 8, 2), (512, 512, 128, 8, 1), (512, 512, 128, 4, 1), (128, 128, 64, 3, 1)):
    gen = torch.Generator().manual_seed(1)
    X, Y = walk(gen, A, M, D).to(dev), walk(gen, B, M, D).to(dev)
    go = torch.randn(A * B, generator=gen, dtype=torch.float64).to(dev)
    K, edges = be.solve_fwd_fused_rbf(X, Y, 1.0, d, False, True, keep_edges=True)
    if be.rbf_adjoint_fused(X, Y, 1.0, d, edges, go) is None:
        print('A=%d B=%d len %d dim %d d=%d: unsupported' % (A, B, M, D, d)); continue
    t_f = tm(lambda: be.rbf_adjoint_fused(X, Y, 1.0, d, edges, go))
    def old():
        inc = be.static_increments(1, 1.0, X, Y, True)
        _, W = be.solve_adj(inc, d, False, edges=edges)
        return be.static_adjoint(1, 1.0, X, Y, W, go, True)
    t_o = tm(old)
    r1 = be.rbf_adjoint_fused(X, Y, 1.0, d, edges, go)
    if r1 is None:
        print('A=%d B=%d len %d dim %d d=%d: unsupported' % (A, B, M, D, d)); continue
    g1 = r1[0]; g2 = old()
    print("A=%d B=%d len %d dim %d d=%d: fused %.2f ms, unfused %.2f ms, rel diff %.2e" % (A, B, M, D, d, t_f, t_o, float((g1 - g2).abs().max() / g2.abs().max())))
