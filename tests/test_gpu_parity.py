"""Parity of the HIP path against the CPU oracle and the reference's golden vectors.  Needs an MI355X.

Everything goes through the C ABI (sigkernel_amd._lib.HipBackend -> libsigkernel_amd.so).
Tolerances: north_star asks for <= 1e-6 relative error in fp64; the simple/exact kernels are
bit-identical to the oracle, the fast kernels (FMA-contracted) are held to 1e-12.
"""
import os

import numpy as np
import pytest
import torch

import sigkernel_amd
from sigkernel_amd import _lib
from conftest import golden, golden_gram_cases, grad_tol, make_kernel, rel_err, walk
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FAST_TOL = 1e-12
ADJ_TOL = 1e-10      # vs the CPU oracle's closed form
F32_RTOL, F32_ATOL = 1e-4, 1e-5   # the reference's own fp32 acceptance (sigkernel/test_mps.py:32)

DEV = "cuda:0"


@pytest.fixture(scope="module")
def be():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    lib = _lib.load()
    assert lib.sk_device_count() >= 1
    return _lib.HipBackend()


SHAPES = [  # P, Mc, Nc, dyadic
    (3, 1, 1, 0), (3, 1, 1, 3), (5, 9, 19, 1), (4, 7, 3, 2), (6, 63, 63, 1), (2, 127, 127, 1),
    (3, 64, 65, 0), (2, 130, 70, 1), (2, 33, 200, 2), (7, 5, 6, 3), (1, 300, 17, 0), (2, 20, 20, 4),
    (700, 9, 19, 1), (3000, 5, 7, 0), (1100, 63, 63, 1), (40, 300, 130, 1), (9, 511, 140, 0), (300, 127, 127, 1),
    (5, 260, 300, 2), (3, 70, 40, 3), (130, 16, 8, 2), (64, 3, 2, 1),
]


def padded(inc, dtype=None):
    """Device copy of inc [P,Mc,Nc] whose rows are 16-byte aligned (the layout sk_increments produces)."""
    t = torch.from_numpy(np.ascontiguousarray(inc))
    if dtype is not None:
        t = t.to(dtype)
    ld = _lib._padded_ld(t.shape[-1], t.element_size())
    buf = torch.zeros(t.shape[:-1] + (ld,), dtype=t.dtype, device=DEV)
    buf[..., : t.shape[-1]] = t.to(DEV)
    return buf[..., : t.shape[-1]]


def _inc(P, Mc, Nc, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return rng.normal(scale=scale, size=(P, Mc, Nc))


@pytest.mark.parametrize("P,Mc,Nc,d", SHAPES)
@pytest.mark.parametrize("naive", [False, True])
def test_forward_exact_kernels_bit_identical_to_oracle(be, P, Mc, Nc, d, naive):
    inc = _inc(P, Mc, Nc, seed=Mc * 1000 + Nc + d)
    want, grid = O.solve_coarse(inc, d, naive, want_grid=True)
    t = torch.from_numpy(inc).to(DEV)
    out, g, e = be.solve_fwd(t, d, naive, flags=_lib.FLAG_EXACT | _lib.FLAG_SIMPLE, want_grid=True, want_edges=True)
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(g.cpu().numpy(), grid)
    edges = np.concatenate([grid[:, -1, :], grid[:, :, -1]], axis=1)
    assert np.array_equal(e.cpu().numpy(), edges)


@pytest.mark.parametrize("P,Mc,Nc,d", SHAPES)
@pytest.mark.parametrize("naive", [False, True])
def test_forward_default_path_matches_oracle(be, P, Mc, Nc, d, naive):
    inc = _inc(P, Mc, Nc, seed=7 + Mc * 1000 + Nc + d)
    want = O.solve_coarse(inc, d, naive)
    out = be.solve_fwd(torch.from_numpy(inc).to(DEV), d, naive)          # dense rows: any layout is accepted
    assert rel_err(out.cpu().numpy(), want) <= FAST_TOL
    if d <= 3:
        out = be.solve_fwd(padded(inc), d, naive, flags=_lib.FLAG_FAST_ONLY)   # the tiled LDS-DMA kernel, no fallback
        assert rel_err(out.cpu().numpy(), want) <= FAST_TOL
        # terminal row/column output: served by the anti-diagonal kernel (exact); the strip kernel keeps its own padded
        # edge layout for the fused adjoint (covered by the adjoint tests: wrong edges blow up the self-check residual)
        _, grid = O.solve_coarse(inc, d, naive, want_grid=True)
        edges = np.concatenate([grid[:, -1, :], grid[:, :, -1]], axis=1)
        out, _, e = be.solve_fwd(padded(inc), d, naive, want_edges=True)
        assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(e.cpu().numpy(), edges)
        with pytest.raises(ValueError):
            be.solve_fwd(padded(inc), d, naive, flags=_lib.FLAG_FAST_ONLY, want_edges=True)


@pytest.mark.parametrize("P,Mc,Nc,d", SHAPES)
def test_forward_fp32_io(be, P, Mc, Nc, d):
    inc = _inc(P, Mc, Nc, seed=11 + Mc + Nc + d).astype(np.float32)
    want = O.solve_coarse(inc.astype(np.float64), d)
    out = be.solve_fwd(torch.from_numpy(inc).to(DEV), d)
    assert out.dtype == torch.float32
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=F32_RTOL, atol=F32_ATOL)
    if d <= 3:
        out = be.solve_fwd(padded(inc), d, flags=_lib.FLAG_FAST_ONLY)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=F32_RTOL, atol=F32_ATOL)


@pytest.mark.parametrize("P,Mc,Nc,d", SHAPES)
@pytest.mark.parametrize("naive", [False, True])
def test_adjoint_matches_oracle(be, P, Mc, Nc, d, naive):
    if P * (Mc << d) * (Nc << d) > 3e8:
        pytest.skip("oracle adjoint too slow for this shape")
    inc = _inc(P, Mc, Nc, seed=3 + Mc * 1000 + Nc + d)
    want_k, want_w = O.adjoint_coarse(inc, d, naive, nthreads=8)
    # stored-grid kernel: bit-identical to the oracle's closed form
    k2, W2 = be.solve_adj(torch.from_numpy(inc).to(DEV), d, naive, flags=_lib.FLAG_EXACT | _lib.FLAG_SIMPLE)
    assert np.array_equal(W2.cpu().numpy(), want_w) and np.array_equal(k2.cpu().numpy(), want_k)
    # default path (fast fused kernel when the shape is covered, else the stored-grid kernel)
    # The fused kernel recomputes K backwards: its error grows like 1e-16 K_max^2 and is tracked by the self-check
    # residual (W error ~ residual / 2); pairs above ADJ_RESIDUAL_TOL = 1e-8 are re-solved exactly.
    k, W, res = be.solve_adj(padded(inc), d, naive, return_residual=True)
    resmax = float(res.max())
    assert rel_err(k.cpu().numpy(), want_k) <= FAST_TOL
    assert rel_err(W.cpu().numpy(), want_w) <= max(ADJ_TOL, 10 * resmax)
    # the residual stays small unless K explodes (long grids with these synthetic increments reach |W| ~ 1e6; every such
    # pair is flagged and re-solved, which the line above has just checked)
    assert resmax <= 1e-2 or np.abs(want_w).max() > 1e4
    fast_ok = 0 <= d <= 2 and (Mc << d) + (Nc << d) + 2 <= 1024
    if fast_ok:
        try:
            k, W, res = be.solve_adj(padded(inc), d, naive, flags=_lib.FLAG_FAST_ONLY, return_residual=True)
        except ValueError:
            return   # shape not covered by the fused kernel (e.g. fewer columns than lanes)
        assert rel_err(W.cpu().numpy(), want_w) <= max(ADJ_TOL, 10 * float(res.max()))
        assert rel_err(k.cpu().numpy(), want_k) <= FAST_TOL


@pytest.mark.parametrize("P,Mc,Nc,d", [(7, 63, 63, 1), (3, 127, 127, 1), (20, 31, 40, 1), (9, 15, 100, 2), (4, 63, 63, 2),
                                        (2, 200, 130, 1), (2, 130, 260, 2), (5, 8, 8, 1), (2, 300, 300, 1), (1, 400, 500, 1),
                                        (6, 63, 63, 0), (3, 255, 255, 0), (2, 300, 70, 0), (9, 20, 500, 0)])
def test_fused_adjoint_without_the_safety_net(be, P, Mc, Nc, d):
    """Tame increments (K stays O(1)): the fused forward(edges) + adjoint kernels alone -- single band, multi-band, partial
    lane groups, grids beyond 1024 nodes per side -- must give W to 1e-10 with a self-check residual at round-off level, so
    nothing here is rescued by the stored-grid re-solve."""
    inc = _inc(P, Mc, Nc, seed=77 + Mc + 3 * Nc + d, scale=0.6 / np.sqrt(Mc * Nc))
    want_k, want_w = O.adjoint_coarse(inc, d, nthreads=8)
    k, W, res = be.solve_adj(padded(inc), d, flags=_lib.FLAG_FAST_ONLY, return_residual=True)
    assert float(res.max()) < 1e-10
    assert rel_err(W.cpu().numpy(), want_w) <= ADJ_TOL and rel_err(k.cpu().numpy(), want_k) <= FAST_TOL
    # fp32 I/O: the fp32 fused adjoint at d = 1, the fp64 one on up-cast increments at d = 2 (never the stored-grid kernel)
    k32, W32, r32 = be.solve_adj(padded(inc.astype(np.float32)), d, flags=0, return_residual=True)
    w32 = O.adjoint_coarse(inc.astype(np.float32).astype(np.float64), d, nthreads=8)[1]
    assert W32.dtype == torch.float32 and float(r32.max()) < 1e-6
    np.testing.assert_allclose(W32.cpu().numpy(), w32, rtol=1e-3, atol=2e-5 * np.abs(w32).max())


@pytest.mark.parametrize("P,Mc,Nc,d", [(9, 63, 63, 1), (3, 200, 130, 1), (5, 40, 70, 2), (4, 31, 31, 1)])
def test_adjoint_from_kept_edges_is_identical(be, P, Mc, Nc, d):
    """sk_solve_fwd_edges + sk_solve_adj(SK_FLAG_EDGES_GIVEN) == the self-contained sk_solve_adj, bit for bit, and K matches."""
    inc = padded(_inc(P, Mc, Nc, seed=5 + Mc + Nc, scale=0.6 / np.sqrt(Mc * Nc)))
    k0, W0, r0 = be.solve_adj(inc, d, flags=_lib.FLAG_FAST_ONLY, return_residual=True)
    k1, edges = be.solve_fwd_keep_edges(inc, d)
    assert edges is not None
    _, W1, r1 = be.solve_adj(inc, d, flags=_lib.FLAG_FAST_ONLY, return_residual=True, edges=edges)
    assert torch.equal(W0, W1) and torch.equal(r0, r1) and torch.equal(k0, k1)
    # shapes outside the strip kernels' adjoint scope: no edges, ordinary forward value
    k2, e2 = be.solve_fwd_keep_edges(inc, 3)
    assert e2 is None and rel_err(k2.cpu().numpy(), O.solve_coarse(inc.cpu().numpy(), 3)) <= FAST_TOL


@pytest.mark.parametrize("A,B,M,N,D,d", [(6, 5, 64, 64, 4, 1), (3, 4, 128, 100, 8, 1), (4, 4, 40, 64, 3, 2), (2, 9, 20, 24, 8, 1)])
def test_fused_linear_forward_keeps_usable_edges(be, A, B, M, N, D, d):
    """The fused linear forward's edges feed the fused adjoint of the (separately formed) increments: same W to 1e-10."""
    gen = torch.Generator().manual_seed(A * 7 + M + N)
    X, Y = (walk(gen, A, M, D) * 2).to(DEV), (walk(gen, B, N, D) * 2).to(DEV)
    K, edges = be.solve_fwd_fused_linear(X, Y, 1.0, d, False, gram=True, keep_edges=True)
    assert edges is not None
    inc = be.static_increments(0, 1.0, X, Y, gram=True)
    k0, W0, r0 = be.solve_adj(inc, d, flags=_lib.FLAG_FAST_ONLY, return_residual=True)
    _, W1, r1 = be.solve_adj(inc, d, flags=_lib.FLAG_FAST_ONLY, return_residual=True, edges=edges)
    assert rel_err(K.cpu().numpy(), k0.cpu().numpy()) <= FAST_TOL
    assert float(r1.max()) <= 1e-9 and rel_err(W1.cpu().numpy(), W0.cpu().numpy()) <= ADJ_TOL


@pytest.mark.parametrize("kind,D,d", [("rbf", 3, 1), ("linear", 8, 1), ("rbf", 4, 2), ("linear", 2, 2), ("linear", 5, 0), ("rbf", 2, 0)])
def test_symmetric_gram_with_gradient_uses_the_triangle(be, kind, D, d, monkeypatch):
    """compute_Gram(X, X, sym=True) with a gradient solves only the blocks on and above the diagonal; values, gradient of a
    non-symmetric loss and the reference's 2x rule must match the full (sym=False) computation."""
    from sigkernel_amd import sigkernel as S
    monkeypatch.setattr(S, "_SYM_TILES", 3)
    monkeypatch.setattr(S, "_SYM_MIN_CELLS", 0.0)
    monkeypatch.setattr(S, "_SYM_MIN_ROWS", 4)
    gen = torch.Generator().manual_seed(17 + D + d)
    X = (walk(gen, 29, 20, D) * 2).to(DEV)
    w = torch.randn(29, 29, generator=gen, dtype=torch.float64).to(DEV)      # NOT symmetric on purpose
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(0.9)
    sk = sigkernel_amd.SigKernel(k, d)
    X1 = X.clone().requires_grad_(True)
    K1 = sk.compute_Gram(X1, X1, sym=True)
    (K1 * w).sum().backward()
    X2 = X.clone().requires_grad_(True)
    K2 = sk.compute_Gram(X2, X2, sym=False)
    (K2 * w).sum().backward()
    assert torch.equal(K1, K1.t())
    assert rel_err(K1.detach().cpu().numpy(), K2.detach().cpu().numpy()) <= 1e-12
    assert rel_err(X1.grad.cpu().numpy(), X2.grad.cpu().numpy()) <= 1e-10


def test_adjoint_self_check_triggers_the_stored_grid_resolve(be):
    """Exploding kernels (|K| ~ 1e9, far outside where the scheme means anything) break the backward recompute of K;
    the residual must flag those pairs and the re-solve must restore the oracle's answer."""
    rng = np.random.default_rng(5)
    inc = rng.normal(scale=0.02, size=(6, 63, 63))
    inc[2] = rng.normal(scale=0.9, size=(63, 63))      # one wild pair among tame ones
    want_k, want_w = O.adjoint_coarse(inc, 1, nthreads=8)
    k, W, res = be.solve_adj(padded(inc), 1, return_residual=True)
    res = res.cpu().numpy()
    assert res[2] > _lib.HipBackend.ADJ_RESIDUAL_TOL and np.all(np.delete(res, 2) < 1e-10)
    assert rel_err(W.cpu().numpy()[2], want_w[2]) <= 1e-12        # re-solved by the bit-exact kernel
    assert rel_err(np.delete(W.cpu().numpy(), 2, axis=0), np.delete(want_w, 2, axis=0)) <= ADJ_TOL


def test_adjoint_of_a_long_first_path_needs_no_rescue_kernel(be):
    """More fine rows than the stored-grid rescue kernel can hold in LDS (three diagonals of MM + 1 doubles > 160 KB, i.e.
    MM > 6826): the multi-band fast adjoint has no such limit, and the unconditional rescue launch must then be skipped,
    not raise `unsupported`."""
    rng = np.random.default_rng(11)
    Mc, Nc, d = 3500, 24, 1                      # MM = 7000
    assert int(_lib.load().sk_adj_rescue_slot_bytes(Mc, Nc, d)) == 0
    inc = rng.normal(scale=0.01, size=(3, Mc, Nc))
    want_k, want_w = O.adjoint_coarse(inc, d, nthreads=8)
    k, W, res = be.solve_adj(padded(inc), d, return_residual=True)
    assert float(res.max()) < 1e-9
    assert rel_err(k.cpu().numpy(), want_k) <= FAST_TOL and rel_err(W.cpu().numpy(), want_w) <= ADJ_TOL


def test_adjoint_rescue_leaves_poisoned_pairs_alone(be):
    """A NaN increment poisons its pair: whatever residual the self-check reports for it (fmax drops NaNs: 0, or NaN), the rescue
    must not spend a stored-grid re-solve on it (the answer is NaN either way) and the other pairs must be untouched."""
    rng = np.random.default_rng(12)
    inc = rng.normal(scale=0.02, size=(5, 40, 40))
    inc[3, 7, 9] = np.nan
    want_k, want_w = O.adjoint_coarse(np.delete(inc, 3, axis=0), 1, nthreads=8)
    k, W, res = be.solve_adj(padded(inc), 1, return_residual=True)
    assert not (res.cpu().numpy()[3] > _lib.HipBackend.ADJ_RESIDUAL_TOL) and np.isnan(W.cpu().numpy()[3]).any()
    assert rel_err(np.delete(W.cpu().numpy(), 3, axis=0), want_w) <= ADJ_TOL


def test_fuzz_random_shapes_against_oracle(be):
    """120 random shapes (ragged, tiny, multi-band, every dyadic order the tiled kernels cover) through every fast kernel."""
    rng = np.random.default_rng(2024)
    lin = sigkernel_amd.LinearKernel()
    for it in range(120):
        d = int(rng.integers(0, 4))
        Mc = int(rng.integers(1, 330 >> (d // 2)))
        Nc = int(rng.integers(1, 330 >> (d // 2)))
        P = int(rng.integers(1, 40))
        naive = bool(rng.integers(0, 2))
        inc = rng.normal(scale=0.03, size=(P, Mc, Nc))
        want = O.solve_coarse(inc, d, naive, nthreads=8)
        got = be.solve_fwd(padded(inc), d, naive, flags=_lib.FLAG_FAST_ONLY)
        assert rel_err(got.cpu().numpy(), want) <= FAST_TOL, (it, P, Mc, Nc, d, naive)
        got32 = be.solve_fwd(padded(inc.astype(np.float32)), d, naive, flags=_lib.FLAG_FAST_ONLY)
        np.testing.assert_allclose(got32.cpu().numpy(), O.solve_coarse(inc.astype(np.float32).astype(np.float64), d, naive),
                                   rtol=F32_RTOL, atol=F32_ATOL)
        if d <= 2 and P * (Mc << d) * (Nc << d) < 4e7:
            wk, ww = O.adjoint_coarse(inc, d, naive, nthreads=8)
            k, W, res = be.solve_adj(padded(inc), d, naive, return_residual=True)
            assert rel_err(W.cpu().numpy(), ww) <= max(ADJ_TOL, 10 * float(res.max())), (it, P, Mc, Nc, d, naive)
            assert rel_err(k.cpu().numpy(), wk) <= FAST_TOL
            # tame increments: nothing may need the stored-grid rescue (a wrong terminal edge would show up here)
            assert float(res.max()) <= 1e-7 * max(1.0, float(np.abs(ww).max())), (it, P, Mc, Nc, d, naive, float(res.max()))
        if d <= 2 and it % 3 == 0:
            A, B, D = int(rng.integers(1, 7)), int(rng.integers(1, 7)), int(rng.integers(1, 9))
            gen = torch.Generator().manual_seed(it)
            Xc, Yc = walk(gen, A, Mc + 1, D) * 2, walk(gen, B, Nc + 1, D) * 2
            K = be.solve_fwd_fused_linear(Xc.to(DEV), Yc.to(DEV), 1.0, d, naive, gram=True)
            if K is not None:
                G = torch.einsum("ipk,jqk->ijpq", Xc, Yc).numpy()
                assert rel_err(K.cpu().numpy(), O.solve_coarse(O.increments(G), d, naive, nthreads=8)) <= 1e-11, (it, A, B, Mc, Nc, D, d)


def test_fused_linear_adjoint_against_the_unfused_route(be, monkeypatch):
    """sk_linear_adjoint_fused_f64 (adjoint PDE + LinearKernel contraction in one kernel, from the paths and the forward's
    edges) against sk_static_increments -> sk_solve_adj -> sk_linear_adjoint on 80 random shapes; dL/dX agrees to 1e-11
    (or to the kernel's own self-check residual where the kernel values are large)."""
    monkeypatch.setattr(type(be), "ADJ_RESIDUAL_TOL", 1e-5)   # compare also where the product path would call the rescue
    rng = np.random.default_rng(99)
    n = 0
    for it in range(80):
        d = int(rng.integers(0, 3))
        cap = 65 if d == 2 else 129      # dyadic 0 sweeps two rows per lane here (four in the forward kernels)
        M = int(rng.integers(2, cap + 1)) if it % 4 else cap
        N = int(rng.integers(2, 200))
        A, B, D = int(rng.integers(1, 30)), int(rng.integers(1, 48)), int(rng.integers(1, 9))
        par = 1.0 if it % 3 else float(rng.uniform(0.5, 1.5))
        gen = torch.Generator().manual_seed(2000 + it)
        X, Y = (walk(gen, A, M, D) * 2).to(DEV), (walk(gen, B, N, D) * 2).to(DEV)
        go = torch.randn(A * B, generator=gen, dtype=torch.float64).to(DEV) if it % 5 else None
        K, edges = be.solve_fwd_fused_linear(X, Y, par, d, False, gram=True, keep_edges=True)
        assert edges is not None
        inc = be.static_increments(0, par, X, Y, gram=True)
        _, W = be.solve_adj(inc, d, False, edges=edges, flags=_lib.FLAG_FAST_ONLY)
        want = be.static_adjoint(0, par, X, Y, W, go, True)
        got = be.linear_adjoint_fused(X, Y, par, d, edges, go)
        assert got is not None, (it, A, B, M, N, D, d)
        n += 1
        assert rel_err(got[0].cpu().numpy(), want.cpu().numpy()) <= max(1e-11, 10 * float(got[1])), (it, A, B, M, N, D, d, par)
    assert n == 80
    # more paths than resident lane groups: one whole Gram row per group, several rounds of workgroups
    gen = torch.Generator().manual_seed(77)
    X, Y = walk(gen, 2600, 6, 2).to(DEV), walk(gen, 3, 7, 2).to(DEV)
    K, edges = be.solve_fwd_fused_linear(X, Y, 1.0, 1, False, gram=True, keep_edges=True)
    inc = be.static_increments(0, 1.0, X, Y, gram=True)
    _, W = be.solve_adj(inc, 1, False, edges=edges)
    got = be.linear_adjoint_fused(X, Y, 1.0, 1, edges, None)
    assert got is not None and float(got[1]) <= 1e-10 and rel_err(got[0].cpu().numpy(), be.static_adjoint(0, 1.0, X, Y, W, None, True).cpu().numpy()) <= 1e-11
    # outside its scope the kernel says so
    X0 = torch.zeros(2, 20, 3, dtype=torch.float64, device=DEV)
    assert be.linear_adjoint_fused(X0, X0, 1.0, 3, torch.zeros(8, dtype=torch.float64, device=DEV), None) is None      # dyadic 3
    X1 = torch.zeros(2, 300, 3, dtype=torch.float64, device=DEV)
    assert be.linear_adjoint_fused(X1, X1, 1.0, 1, torch.zeros(8, dtype=torch.float64, device=DEV), None) is None      # two bands


@pytest.mark.parametrize("A,M,N,D,d,par", [(7, 40, 33, 5, 1, 1.0), (70, 64, 64, 8, 2, 0.8), (3, 128, 17, 2, 1, 1.4), (1, 2, 2, 1, 2, 1.0),
                                              (9, 129, 50, 8, 0, 1.0), (5, 30, 130, 3, 0, 0.9)])
def test_fused_linear_adjoint_paired_and_fp32(be, A, M, N, D, d, par, monkeypatch):
    """Paired batches (compute_kernel gradients) and fp32 inputs (swept in fp64) through the fused adjoint."""
    gen = torch.Generator().manual_seed(A + M + N)
    X, Y = (walk(gen, A, M, D) * 2).to(DEV), (walk(gen, A, N, D) * 2).to(DEV)
    go = torch.randn(A, generator=gen, dtype=torch.float64).to(DEV)
    K, edges = be.solve_fwd_fused_linear(X, Y, par, d, False, gram=False, keep_edges=True)
    assert edges is not None
    inc = be.static_increments(0, par, X, Y, gram=False)
    _, W = be.solve_adj(inc, d, False, edges=edges)
    want = be.static_adjoint(0, par, X, Y, W, go, False)
    got = be.linear_adjoint_fused(X, Y, par, d, edges, go, gram=False)
    assert got is not None and rel_err(got[0].cpu().numpy(), want.cpu().numpy()) <= 1e-11
    _, edges32 = be.solve_fwd_fused_linear(X.float().double(), Y.float().double(), par, d, False, gram=False, keep_edges=True)
    got32 = be.linear_adjoint_fused(X.float(), Y.float(), par, d, edges32, go.float(), gram=False)
    assert got32 is not None and got32[0].dtype == torch.float32
    got32 = got32[0]
    np.testing.assert_allclose(got32.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(want.abs().max()))
    # API level: compute_kernel gradients, fp64 and fp32, against the unfused route
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(scale=par), d)
    for dt, tol in ((torch.float64, 1e-10), (torch.float32, 2e-4)):
        res = []
        for env in ("", "1"):
            monkeypatch.setattr(sigkernel_amd.routes, "no_fused_adjoint", bool(env))
            Xg = X.to(dt).clone().requires_grad_(True)
            (sk.compute_kernel(Xg, Y.to(dt)) * go.to(dt)).sum().backward()
            res.append(Xg.grad.double().cpu().numpy())
        monkeypatch.setattr(sigkernel_amd.routes, "no_fused_adjoint", False)
        assert rel_err(res[0], res[1]) <= tol, (dt, rel_err(res[0], res[1]))


def test_fused_linear_adjoint_is_what_the_api_runs(be, monkeypatch):
    """compute_mmd / compute_Gram gradients with LinearKernel go through the fused adjoint and agree with the unfused route."""
    gen = torch.Generator().manual_seed(31)
    X, Y = (walk(gen, 12, 40, 5)).to(DEV), (walk(gen, 9, 33, 5)).to(DEV)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1)
    calls = []
    orig = type(be).linear_adjoint_fused
    monkeypatch.setattr(type(be), "linear_adjoint_fused", lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1])
    X1 = X.clone().requires_grad_(True)
    sk.compute_mmd(X1, Y).backward()
    assert calls, "LinearKernel backward did not use the fused adjoint"
    monkeypatch.setattr(sigkernel_amd.routes, "no_fused_adjoint", True)
    X2 = X.clone().requires_grad_(True)
    sk.compute_mmd(X2, Y).backward()
    assert rel_err(X1.grad.cpu().numpy(), X2.grad.cpu().numpy()) <= 1e-10


def test_fuzz_fused_forwards_against_the_streaming_route(be):
    """150 random shapes: the fused linear / RBF forwards (values, and the edges they keep for the adjoint) against
    sk_static_increments + sk_solve_fwd / sk_solve_adj on the same paths -- GPU against GPU, so large batches are cheap."""
    rng = np.random.default_rng(77)
    n_fused = n_edges = 0
    for it in range(150):
        d = int(rng.integers(0, 3))
        cap = 64 * (4 >> d)
        M = int(rng.integers(2, cap + 1)) if it % 5 else cap            # the largest single-band length every fifth time
        N = int(rng.integers(2, 300))
        A, B, D = int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 9))
        naive = bool(rng.integers(0, 2))
        kind = int(rng.integers(0, 2))
        gen = torch.Generator().manual_seed(1000 + it)
        X, Y = (walk(gen, A, M, D) * 2).to(DEV), (walk(gen, B, N, D) * 2).to(DEV)
        param = 1.0 if kind == 0 else float(rng.uniform(0.3, 2.0))
        inc = be.static_increments(kind, param, X, Y, gram=True)
        want = be.solve_fwd(inc, d, naive)
        if kind == 0:
            res = be.solve_fwd_fused_linear(X, Y, 1.0, d, naive, gram=True, keep_edges=True)
        else:
            res = be.solve_fwd_fused_rbf(X, Y, param, d, naive, gram=True, keep_edges=True)
        if res is None:      # round 3: no RBF variant at dyadic 0 for the naive scheme / dims > 4 (see test_fused_rbf_forward_matches_oracle)
            assert kind == 1 and d == 0 and (naive or D > 4), (it, kind, A, B, M, N, D, d)
            continue
        K, edges = res
        n_fused += 1
        assert rel_err(K.cpu().numpy(), want.cpu().numpy()) <= 1e-12, (it, kind, A, B, M, N, D, d, naive)
        if edges is not None and A * B * (M << d) * (N << d) < 6e7:
            n_edges += 1
            _, W0, r0 = be.solve_adj(inc, d, naive, flags=_lib.FLAG_FAST_ONLY, return_residual=True)
            _, W1, r1 = be.solve_adj(inc, d, naive, flags=_lib.FLAG_FAST_ONLY, return_residual=True, edges=edges)
            tol = max(ADJ_TOL, 10 * float(r0.max()))
            assert rel_err(W1.cpu().numpy(), W0.cpu().numpy()) <= tol, (it, kind, A, B, M, N, D, d, naive)
    assert n_fused >= 120 and n_edges >= 50


def test_increments_and_transpose_bit_identical(be):
    rng = np.random.default_rng(0)
    for shape in [(3, 2, 2), (4, 10, 20), (2, 3, 128, 128), (5, 65, 7), (1, 300, 300)]:
        G = rng.normal(size=shape)
        got = be.increments(torch.from_numpy(G).to(DEV)).cpu().numpy()
        assert np.array_equal(got, O.increments(G))
        W = rng.normal(size=shape[:-2] + (shape[-2] - 1, shape[-1] - 1))
        got = be.increments_adjoint(torch.from_numpy(W).to(DEV)).cpu().numpy()
        assert np.array_equal(got, O.increments_adjoint(W))
        s = rng.normal(size=shape[:-2])
        got = be.increments_adjoint(torch.from_numpy(W).to(DEV), torch.from_numpy(s).to(DEV)).cpu().numpy()
        assert rel_err(got, O.increments_adjoint(W) * s[..., None, None]) <= 1e-15
    G32 = rng.normal(size=(3, 9, 11)).astype(np.float32)
    got = be.increments(torch.from_numpy(G32).to(DEV)).cpu().numpy()
    want = ((G32[:, 1:, 1:] + G32[:, :-1, :-1]) - G32[:, 1:, :-1]) - G32[:, :-1, 1:]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kind", ["linear", "rbf"])
@pytest.mark.parametrize("A,B,M,N,D", [(3, 4, 10, 20, 2), (2, 3, 128, 128, 8), (5, 2, 64, 64, 4), (2, 2, 70, 131, 17),
                                       (1, 1, 2, 2, 1), (3, 3, 33, 64, 32)])
def test_fused_static_increments_match_the_generic_route(be, kind, A, B, M, N, D):
    gen = torch.Generator().manual_seed(A * 100 + M + N + D)
    for dtype, tol in ((torch.float64, 2e-15), (torch.float32, 2e-6)):
        X = walk(gen, A, M, D).to(dtype).to(DEV) * 3
        Y = walk(gen, B, N, D).to(dtype).to(DEV) * 3
        for scale in ((1.0, 0.7) if kind == "linear" else (0.5, 2.0)):
            k = sigkernel_amd.LinearKernel(scale) if kind == "linear" else sigkernel_amd.RBFKernel(scale)
            code, param = (0, scale) if kind == "linear" else (1, scale)
            # Gram: the reference's linear Gram_matrix ignores `scale` -> param 1
            G = k.Gram_matrix(X.double(), Y.double())
            want = be.increments(G.contiguous()).cpu().numpy()
            got = be.static_increments(code, 1.0 if kind == "linear" else param, X, Y, gram=True)
            assert got.shape == (A, B, M - 1, N - 1) and got.stride(-2) * got.element_size() % 128 == 0
            assert np.max(np.abs(got.double().cpu().numpy() - want)) <= tol * max(1.0, float(G.abs().max()))
            # zero padding behind the view (the adjoint sweep and the edge output rely on it)
            base = got.as_strided((A, B, M - 1, got.stride(-2)), got.stride())
            assert torch.all(base[..., N - 1:] == 0)
            # paired
            n = min(A, B)
            Gp = k.batch_kernel(X[:n].double(), Y[:n].double())
            wantp = be.increments(Gp.contiguous()).cpu().numpy()
            gotp = be.static_increments(code, param, X[:n].contiguous(), Y[:n].contiguous(), gram=False)
            assert np.max(np.abs(gotp.double().cpu().numpy() - wantp)) <= tol * max(1.0, float(Gp.abs().max()))


@pytest.mark.parametrize("kind", ["linear", "rbf"])
@pytest.mark.parametrize("A,B,M,N,D", [(3, 4, 10, 20, 2), (2, 3, 128, 128, 8), (5, 2, 64, 64, 4), (2, 2, 70, 131, 17), (1, 1, 2, 2, 1),
                                       # 9..32 dims, at most 128 points of y: the tiled contractions (rows per block 16 / 8; both column passes)
                                       (3, 5, 70, 128, 17), (4, 3, 9, 66, 9), (2, 3, 40, 30, 32), (3, 2, 21, 2, 24), (2, 4, 2, 65, 12)])
def test_fused_static_adjoint_matches_autograd_through_the_static_kernel(be, kind, A, B, M, N, D):
    gen = torch.Generator().manual_seed(7 + A * 100 + M + N + D)
    X = (walk(gen, A, M, D) * 3).to(DEV)
    Y = (walk(gen, B, N, D) * 3).to(DEV)
    for scale in ((1.0, 0.7) if kind == "linear" else (0.5, 2.0)):
        k = sigkernel_amd.LinearKernel(scale) if kind == "linear" else sigkernel_amd.RBFKernel(scale)
        code = 0 if kind == "linear" else 1
        for gram in (True, False):
            Xt, Yt = (X, Y) if gram else (X[:min(A, B)].contiguous(), Y[:min(A, B)].contiguous())
            shape = (Xt.shape[0], Yt.shape[0], M - 1, N - 1) if gram else (Xt.shape[0], M - 1, N - 1)
            W = padded(np.random.default_rng(3).normal(size=shape))
            go = torch.randn(shape[:-2], generator=gen, dtype=torch.float64).to(DEV)
            Xg = Xt.clone().requires_grad_(True)
            G = k.Gram_matrix(Xg, Yt) if gram else k.batch_kernel(Xg, Yt)
            (want,) = torch.autograd.grad(G, Xg, be.increments_adjoint(W, go))
            param = (1.0 if gram else scale) if kind == "linear" else scale
            got = be.static_adjoint(code, param, Xt, Yt, W, go, gram)
            assert rel_err(got.cpu().numpy(), want.cpu().numpy()) <= 1e-12


@pytest.mark.parametrize("A,B,M,N,D", [(3, 5, 40, 128, 17), (4, 3, 9, 64, 9), (2, 3, 40, 30, 32), (5, 4, 33, 100, 30), (2, 6, 18, 65, 16)])
def test_linear_static_adjoint_of_wide_paths_in_fp32_and_against_the_batched_product(be, A, B, M, N, D):
    """sk_static_adjoint_* kind 0 (k_static_linear_adj_tiled, 9..32 dims): against the batched matrix product it replaced, written out in
    torch fp64 -- the fp64 entry to rounding, the fp32 one to fp32 rounding of its inputs; all pairs and paired."""
    gen = torch.Generator().manual_seed(11 + A + M + N + D)
    X, Y = walk(gen, A, M, D).to(DEV), (walk(gen, B, N, D) * 2).to(DEV)
    for gram in (True, False):
        Xt, Yt = (X, Y) if gram else (X[:min(A, B)].contiguous(), Y[:min(A, B)].contiguous())
        shape = (Xt.shape[0], Yt.shape[0], M - 1, N - 1) if gram else (Xt.shape[0], M - 1, N - 1)
        W = padded(np.random.default_rng(5).normal(size=shape))
        go = torch.randn(shape[:-2], generator=gen, dtype=torch.float64).to(DEV)
        dY = Yt[:, 1:] - Yt[:, :-1]
        T = (torch.matmul(W, dY) * go[..., None, None]).sum(1) if gram else torch.matmul(W, dY) * go[:, None, None]
        want = torch.zeros_like(Xt)
        want[:, 1:] += T
        want[:, :-1] -= T
        want *= 0.7 ** 2
        got = be.static_adjoint(0, 0.7, Xt, Yt, W, go, gram)
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) <= 1e-12
        W32 = padded(W.cpu().numpy().astype(np.float32))
        got32 = be.static_adjoint(0, 0.7, Xt.float(), Yt.float(), W32, go.float(), gram)
        assert got32.dtype == torch.float32 and rel_err(got32.double().cpu().numpy(), want.cpu().numpy()) <= 2e-6
        assert be.static_adjoint(0, 0.7, Xt, Yt, W, None, gram).shape == Xt.shape      # (no upstream gradient: scale 1)


@pytest.mark.parametrize("kind", ["linear", "rbf"])
@pytest.mark.parametrize("A,B,M,N,D,b0", [(3, 4, 10, 20, 2, 0), (2, 5, 64, 70, 8, 2), (5, 3, 33, 64, 4, 1), (2, 2, 9, 131, 17, 0), (1, 1, 2, 2, 1, 0)])
def test_second_argument_static_adjoint_matches_autograd(be, kind, A, B, M, N, D, b0):
    """sk_static_adjoint2_*: dL/dY for the pairs (a, b >= b0), against torch autograd through the static kernel."""
    gen = torch.Generator().manual_seed(11 + A * 100 + M + N + D)
    X = (walk(gen, A, M, D) * 3).to(DEV)
    Y = (walk(gen, B, N, D) * 3).to(DEV)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(0.7)
    code, param = (0, 1.0) if kind == "linear" else (1, 0.7)
    W = padded(np.random.default_rng(4).normal(size=(A, B, M - 1, N - 1)))
    go = torch.randn(A, B, generator=gen, dtype=torch.float64).to(DEV)
    got = be.static_adjoint2(code, param, X, Y, W, go, b0)
    if kind == "linear" and D > 8:
        assert got is None
        return
    Yg = Y.clone().requires_grad_(True)
    G = k.Gram_matrix(X, Yg)
    (want,) = torch.autograd.grad(G, Yg, be.increments_adjoint(W, go))
    # pairs with b < b0 are excluded: compare rows b >= b0 of a run that zeroes their upstream gradient
    go0 = go.clone()
    go0[:, :b0] = 0
    (want0,) = torch.autograd.grad(k.Gram_matrix(X, Yg), Yg, be.increments_adjoint(W, go0))
    assert got.shape == (B - b0, N, D)
    assert rel_err(got.cpu().numpy(), want0[b0:].cpu().numpy()) <= 1e-12
    if b0 == 0:
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) <= 1e-12


@pytest.mark.parametrize("A,B,M,N,D,d", [(3, 4, 10, 20, 2, 1), (2, 3, 128, 128, 8, 1), (70, 9, 64, 64, 4, 2), (5, 130, 33, 17, 3, 0),
                                            (1, 1, 2, 2, 1, 0), (9, 7, 128, 40, 8, 1), (4, 4, 65, 128, 5, 2), (300, 300, 20, 24, 8, 1),
                                            (2, 2, 257, 30, 6, 0)])
@pytest.mark.parametrize("naive", [False, True])
def test_fused_linear_forward_matches_oracle(be, A, B, M, N, D, d, naive):
    gen = torch.Generator().manual_seed(A + B + M + N + D + d)
    Xc, Yc = walk(gen, A, M, D) * 2, walk(gen, B, N, D) * 2
    X, Y = Xc.to(DEV), Yc.to(DEV)
    K = be.solve_fwd_fused_linear(X, Y, 1.0, d, naive, gram=True)
    assert K is not None, "shape is inside the fused kernel's scope"
    G = torch.einsum("ipk,jqk->ijpq", Xc, Yc).numpy()
    want = O.solve_coarse(O.increments(G), d, naive, nthreads=8)
    assert rel_err(K.cpu().numpy(), want) <= 1e-11
    n = min(A, B)
    Kp = be.solve_fwd_fused_linear(X[:n].contiguous(), Y[:n].contiguous(), 0.8, d, naive, gram=False)
    Gp = torch.bmm(0.8 * Xc[:n], (0.8 * Yc[:n]).transpose(1, 2)).numpy()
    assert rel_err(Kp.cpu().numpy(), O.solve_coarse(O.increments(Gp), d, naive, nthreads=8)) <= 1e-11
    K32 = be.solve_fwd_fused_linear(X.float(), Y.float(), 1.0, d, naive, gram=True)
    np.testing.assert_allclose(K32.cpu().numpy(), want, rtol=1e-4, atol=1e-5)


def test_fused_linear_forward_scope(be):
    X = torch.zeros(2, 300, 3, dtype=torch.float64, device=DEV)
    assert be.solve_fwd_fused_linear(X, X, 1.0, 1, False, True) is None          # 299 coarse rows > 128: two bands
    Xs = X[:, :20].contiguous()
    assert be.solve_fwd_fused_linear(Xs, Xs, 1.0, 3, False, True) is None      # dyadic 3
    X9 = torch.zeros(2, 20, 9, dtype=torch.float64, device=DEV)
    assert be.solve_fwd_fused_linear(X9, X9, 1.0, 1, False, True) is None        # dim 9


@pytest.mark.parametrize("A,B,M,N,D,d", [(3, 4, 10, 20, 2, 1), (2, 3, 128, 128, 8, 1), (70, 9, 64, 64, 4, 2), (5, 130, 33, 17, 3, 0),
                                            (1, 1, 2, 2, 1, 0), (9, 7, 128, 41, 8, 1), (4, 4, 64, 127, 5, 2), (300, 300, 20, 24, 8, 1),
                                            (2, 2, 256, 30, 6, 0), (3, 3, 127, 129, 7, 1), (5, 5, 17, 65, 1, 2)])
@pytest.mark.parametrize("naive", [False, True])
def test_fused_rbf_forward_matches_oracle(be, A, B, M, N, D, d, naive):
    """sk_solve_fwd_rbf_*: nodes, increments and PDE in one kernel, against the oracle's RBF Gram (odd and even lengths: the
    node stream of a pair is one column longer than its increments; M up to the last lane's padding row)."""
    gen = torch.Generator().manual_seed(A + B + M + N + D + d)
    Xc, Yc = walk(gen, A, M, D) * 2, walk(gen, B, N, D) * 2
    X, Y = Xc.to(DEV), Yc.to(DEV)
    sigma = 0.7
    want = O.gram_forward(Xc, Yc, sigkernel_amd.RBFKernel(sigma), d, naive=naive, nthreads=8)
    # at dyadic 0 the four-rows-per-lane kernel is built for the 4-dim fp64 default-scheme case (the 8-dim ones need 280-290
    # registers and spill); everything else takes the two-rows-per-lane variant (round 4): up to 128 node rows, else `unsupported`
    # and the API falls back
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(sigma), d, _naive_solver=naive)
    K = be.solve_fwd_fused_rbf(X, Y, sigma, d, naive, gram=True)
    if d == 0 and (naive or D > 4):
        assert (K is None) == (M > 128)
        if K is None:
            K = sk.compute_Gram(X, Y)
    assert K is not None, "shape is inside the fused kernel's scope"
    assert rel_err(K.cpu().numpy(), want) <= 1e-12
    n = min(A, B)
    Kp = be.solve_fwd_fused_rbf(X[:n].contiguous(), Y[:n].contiguous(), sigma, d, naive, gram=False)
    if Kp is None:
        assert d == 0
        Kp = sk.compute_kernel(X[:n].contiguous(), Y[:n].contiguous())
    assert rel_err(Kp.cpu().numpy(), np.diag(want)[:n]) <= 1e-12
    K32 = be.solve_fwd_fused_rbf(X.float(), Y.float(), sigma, d, naive, gram=True)
    if K32 is None:
        assert d == 0
        K32 = sk.compute_Gram(X.float(), Y.float())
    np.testing.assert_allclose(K32.cpu().numpy(), want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("A,B,M,N,D,d", [(6, 5, 64, 64, 4, 1), (3, 4, 128, 100, 8, 1), (4, 4, 33, 65, 3, 2), (2, 9, 20, 24, 8, 1),
                                            (3, 3, 17, 129, 2, 0), (5, 4, 64, 17, 4, 2)])
def test_fused_rbf_forward_keeps_usable_edges(be, A, B, M, N, D, d):
    """The fused RBF forward's edges (written in the adjoint's strip layout, whose lane / unit counts can be smaller than
    the node stream's) feed the fused adjoint of the separately formed increments: same W to 1e-10."""
    gen = torch.Generator().manual_seed(A * 7 + M + N)
    X, Y = (walk(gen, A, M, D) * 2).to(DEV), (walk(gen, B, N, D) * 2).to(DEV)
    K, edges = be.solve_fwd_fused_rbf(X, Y, 0.8, d, False, gram=True, keep_edges=True)
    assert edges is not None
    inc = be.static_increments(1, 0.8, X, Y, gram=True)
    k0, W0, r0 = be.solve_adj(inc, d, flags=_lib.FLAG_FAST_ONLY, return_residual=True)
    _, W1, r1 = be.solve_adj(inc, d, flags=_lib.FLAG_FAST_ONLY, return_residual=True, edges=edges)
    assert rel_err(K.cpu().numpy(), k0.cpu().numpy()) <= FAST_TOL
    assert float(r1.max()) <= 1e-9 and rel_err(W1.cpu().numpy(), W0.cpu().numpy()) <= ADJ_TOL


def test_fused_rbf_forward_far_apart_points(be):
    """Points so far apart that every node underflows (exp of -1e6 and of -1e300): all increments vanish, K = 1 exactly."""
    X = torch.zeros(2, 12, 3, dtype=torch.float64, device=DEV)
    for shift in (1e3, 1e150):
        Y = X + shift
        K = be.solve_fwd_fused_rbf(X, Y, 1.0, 1, False, gram=True)
        assert torch.equal(K, torch.ones_like(K))
    # a mix: one path of Y coincides with X, the other is far away
    gen = torch.Generator().manual_seed(3)
    Xw = walk(gen, 2, 12, 3).to(DEV)
    Yw = torch.stack([Xw[0], Xw[1] + 1e8])
    K = be.solve_fwd_fused_rbf(Xw, Yw, 0.5, 1, False, gram=True)
    want = O.gram_forward(Xw.cpu(), Yw.cpu(), sigkernel_amd.RBFKernel(0.5), 1)
    assert rel_err(K.cpu().numpy(), want) <= 1e-12


def test_fused_rbf_forward_scope_and_route(be, monkeypatch):
    X = torch.zeros(2, 300, 3, dtype=torch.float64, device=DEV)
    assert be.solve_fwd_fused_rbf(X, X, 1.0, 1, False, True) is None            # 300 node rows > 128: two bands
    Xs = X[:, :20].contiguous()
    assert be.solve_fwd_fused_rbf(Xs, Xs, 1.0, 3, False, True) is None          # dyadic 3
    X9 = torch.zeros(2, 20, 9, dtype=torch.float64, device=DEV)
    assert be.solve_fwd_fused_rbf(X9, X9, 1.0, 1, False, True) is None          # dim 9
    X65 = torch.zeros(2, 65, 2, dtype=torch.float64, device=DEV)
    assert be.solve_fwd_fused_rbf(X65, X65, 1.0, 2, False, True) is None        # 65 node rows at dyadic 2: one more than 64 lanes
    # compute_Gram without a gradient takes the fused kernel and agrees with the increments-in-HBM route
    gen = torch.Generator().manual_seed(5)
    Xw, Yw = walk(gen, 9, 50, 3).to(DEV), walk(gen, 7, 41, 3).to(DEV)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.6), 1)
    calls = []
    orig = type(be).solve_fwd_fused_rbf
    monkeypatch.setattr(type(be), "solve_fwd_fused_rbf", lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1])
    K1 = sk.compute_Gram(Xw, Yw)
    assert calls, "RBFKernel forward without gradient did not use the fused kernel"
    monkeypatch.setattr(sigkernel_amd.routes, "no_fused_rbf", True)
    K2 = sk.compute_Gram(Xw, Yw)
    assert rel_err(K1.cpu().numpy(), K2.cpu().numpy()) <= 1e-12


class _SubclassedLinear(sigkernel_amd.LinearKernel):
    """Not `type(...) is LinearKernel`: must take the generic Gram_matrix / autograd route."""


def test_generic_static_kernel_route_agrees_with_the_fused_one(be):
    gen = torch.Generator().manual_seed(11)
    X, Y = walk(gen, 6, 40, 3).to(DEV), walk(gen, 5, 33, 3).to(DEV)
    w = torch.randn(6, 5, generator=gen, dtype=torch.float64).to(DEV)
    res = []
    for k in (sigkernel_amd.LinearKernel(), _SubclassedLinear()):
        sk = sigkernel_amd.SigKernel(k, 1)
        Xg = X.clone().requires_grad_(True)
        K = sk.compute_Gram(Xg, Y)
        (K * w).sum().backward()
        res.append((K.detach().cpu().numpy(), Xg.grad.cpu().numpy()))
    assert rel_err(res[0][0], res[1][0]) <= 1e-12 and rel_err(res[0][1], res[1][1]) <= 1e-10


# ---------------------------------------------------------------------------------------------
# API level, against the golden vectors produced by the real reference
# ---------------------------------------------------------------------------------------------
def _sk(c, **kw):
    return sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"]), _naive_solver=bool(c["naive"]), **kw)


@pytest.mark.parametrize("name", golden_gram_cases())
def test_api_gram_and_gradients_vs_reference(be, name):
    c = golden(name)
    X, Y, w = (torch.from_numpy(c[k]).to(DEV) for k in ("X", "Y", "w"))
    sk = _sk(c)
    K = sk.compute_Gram(X, Y)
    assert K.device == X.device and K.dtype == X.dtype
    assert rel_err(K.cpu().numpy(), c["gram"]) <= 1e-11
    Xg = X.clone().requires_grad_(True)
    (sk.compute_Gram(Xg, Y) * w).sum().backward()
    assert rel_err(Xg.grad.cpu().numpy(), c["grad_w"]) <= grad_tol(name, "grad_w")
    n = c["paired"].shape[0]
    Xg = X[:n].clone().requires_grad_(True)
    Kp = sk.compute_kernel(Xg, Y[:n])
    assert rel_err(Kp.detach().cpu().numpy(), c["paired"]) <= 1e-11
    (Kp * torch.from_numpy(c["wp"]).to(DEV)).sum().backward()
    assert rel_err(Xg.grad.cpu().numpy(), c["grad_paired"]) <= grad_tol(name, "grad_paired")
    if "mmd" in c:
        Xg = X.clone().requires_grad_(True)
        mmd = sk.compute_mmd(Xg, Y)
        mmd.backward()
        assert abs(float(mmd.detach()) - float(c["mmd"])) <= 1e-11
        assert rel_err(Xg.grad.cpu().numpy(), c["grad_mmd"]) <= grad_tol(name, "grad_mmd")
        Xg = X.clone().requires_grad_(True)
        G = sk.compute_Gram(Xg, Xg, sym=True)
        G.sum().backward()
        assert rel_err(G.detach().cpu().numpy(), c["gram_xx_sym"]) <= 1e-11
        assert rel_err(Xg.grad.cpu().numpy(), c["grad_xx_sum"]) <= grad_tol(name, "grad_xx_sum")


def test_api_readme_example(be):
    c = golden("readme_c1")
    X, Y, Z = (torch.from_numpy(c[k]).to(DEV) for k in ("X", "Y", "Z"))
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(sigma=0.5), dyadic_order=1)
    assert rel_err(sk.compute_kernel(X, Y).cpu().numpy(), c["kernel"]) <= 1e-11
    assert rel_err(sk.compute_Gram(X, Y, sym=False).cpu().numpy(), c["gram"]) <= 1e-11
    Xg = X.clone().requires_grad_(True)
    mmd = sk.compute_mmd(Xg, Y)
    mmd.backward()
    assert abs(float(mmd.detach()) - float(c["mmd"])) <= 1e-11
    assert rel_err(Xg.grad.cpu().numpy(), c["grad_mmd"]) <= grad_tol("readme_c1", "grad_mmd")
    assert abs(float(sk.compute_scoring_rule(X, Z[:1])) - float(c["scoring_rule"])) <= 1e-11
    assert abs(float(sk.compute_expected_scoring_rule(X, Z)) - float(c["expected_scoring_rule"])) <= 1e-11
    assert abs(float(sk.compute_distance(X, Y)) - float(c["distance"])) <= 1e-11


@pytest.mark.parametrize("name", ["gram_c2mini_rbf_d1", "gram_c3mini_lin_d1", "gram_c4mini_rbf_d2", "gram_lin_d0_ragged"])
def test_api_float32_tensors(be, name):
    """The reference's CPU path rejects float32 (cython_backend.pyx:7,64); ours accepts it (fp32 I/O, fp64 PDE state) and
    is held to the reference's own fp32 bar (test_mps.py:32) against the fp64 fixtures."""
    c = golden(name)
    X, Y, w = (torch.from_numpy(c[k]).float().to(DEV) for k in ("X", "Y", "w"))
    sk = _sk(c)
    Xg = X.clone().requires_grad_(True)
    K = sk.compute_Gram(Xg, Y)
    assert K.dtype == torch.float32
    np.testing.assert_allclose(K.detach().cpu().numpy(), c["gram"], rtol=F32_RTOL, atol=F32_ATOL)
    (K * w).sum().backward()
    assert Xg.grad.dtype == torch.float32
    assert rel_err(Xg.grad.cpu().numpy(), c["grad_w"]) <= 2e-4
    n = c["paired"].shape[0]
    np.testing.assert_allclose(sk.compute_kernel(X[:n], Y[:n]).cpu().numpy(), c["paired"], rtol=F32_RTOL, atol=F32_ATOL)


def test_api_rejects_other_dtypes_and_cpu(be):
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 0)
    X = torch.rand(2, 5, 2, device=DEV).half()
    with pytest.raises((TypeError, RuntimeError)):
        sk.compute_Gram(X, X)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sk.compute_Gram(torch.rand(2, 5, 2, dtype=torch.float64), torch.rand(2, 5, 2, dtype=torch.float64))


@pytest.mark.parametrize("kern", ["linear", "rbf"])
@pytest.mark.parametrize("A,M", [(5, 40), (64, 40), (150, 40), (256, 150)])
def test_symmetric_gram_shortcut(be, kern, A, M):
    """sym=True on X against itself without a gradient solves only the blocks on/above the diagonal (large problems)."""
    gen = torch.Generator().manual_seed(A)
    X = walk(gen, A, M, 3).to(DEV)
    k = sigkernel_amd.LinearKernel() if kern == "linear" else sigkernel_amd.RBFKernel(1.0)
    sk = sigkernel_amd.SigKernel(k, 1)
    Ks = sk.compute_Gram(X, X, sym=True)
    Kf = sk.compute_Gram(X, X, sym=False)
    assert torch.equal(Ks, Ks.t())                               # exactly symmetric, like the reference's sym=True
    assert rel_err(Ks.cpu().numpy(), Kf.cpu().numpy()) <= 1e-12
    # with a gradient pending the full matrix is solved and the 2x rule applies (fixtures cover its values)
    Xg = X.clone().requires_grad_(True)
    Kg = sk.compute_Gram(Xg, Xg, sym=True)
    assert rel_err(Kg.detach().cpu().numpy(), Kf.cpu().numpy()) <= 1e-12
    Kg.sum().backward()
    assert Xg.grad is not None and torch.isfinite(Xg.grad).all()
    # different tensors with sym=True: no shortcut, plain result
    Y = walk(gen, A, M, 3).to(DEV)
    assert rel_err(sk.compute_Gram(X, Y, sym=True).cpu().numpy(), sk.compute_Gram(X, Y).cpu().numpy()) <= 1e-15


def test_api_tiling_independence_on_device(be):
    c = golden("gram_c2mini_rbf_d1")
    X, Y = torch.from_numpy(c["X"]).to(DEV), torch.from_numpy(c["Y"]).to(DEV)
    a = _sk(c).compute_Gram(X, Y)
    b = _sk(c, workspace_bytes=1).compute_Gram(X, Y, max_batch=2)
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------
# BASELINE-size inputs: size-independent properties + spot checks against the oracle
# ---------------------------------------------------------------------------------------------
def test_headline_size_properties(be):
    """C3-sized problems (len 128, dim 8, Linear, d=1): a 96 x 96 block of the headline Gram.
    (the full 512 x 512 is run by bench.py with the same spot check)"""
    gen = torch.Generator().manual_seed(0)
    A = B = 96
    X = walk(gen, A, 128, 8).to(DEV)
    Y = walk(gen, B, 128, 8).to(DEV)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1)
    K = sk.compute_Gram(X, Y)
    # (1) spot check 64 random pairs against the oracle
    rng = np.random.default_rng(0)
    idx = rng.integers(0, A * B, size=64)
    Xc, Yc = X.cpu(), Y.cpu()
    for p in idx:
        a, b = divmod(int(p), B)
        want = O.gram_forward(Xc[a:a + 1], Yc[b:b + 1], sigkernel_amd.LinearKernel(), 1)[0, 0]
        assert abs(float(K[a, b]) - want) <= 1e-11 * abs(want)
    # (2) symmetry k(x,y) = k(y,x)
    Kt = sk.compute_Gram(Y, X)
    assert rel_err(Kt.t().cpu().numpy(), K.cpu().numpy()) <= 1e-12
    # (3) a constant path has zero increments: k = 1 exactly; translating x leaves a linear-kernel Gram unchanged
    Yconst = Y[:, :1, :].expand(-1, 128, -1).contiguous()
    assert torch.all((sk.compute_Gram(X, Yconst) - 1.0).abs() <= 1e-10)   # increments are 0 up to GEMM rounding
    Ks = sk.compute_Gram(X + 0.25, Y)
    assert rel_err(Ks.cpu().numpy(), K.cpu().numpy()) <= 1e-9
    # (4) invariance under a permutation of the batch
    perm = torch.randperm(A, generator=gen).to(DEV)
    assert torch.equal(sk.compute_Gram(X[perm].contiguous(), Y), K[perm])
    # (5) paired kernel is the Gram diagonal
    assert rel_err(sk.compute_kernel(X, Y).cpu().numpy(), torch.diagonal(K).cpu().numpy()) <= 1e-12


def test_straight_line_known_answer_in_a_batch(be):
    """Two straight lines with <dx,dy> = 1: one step at d=0 gives 2.25 (exactly in the reference's operand
    order -- the exact kernels reproduce that bit for bit elsewhere; the fused kernel's three-operation coefficients on the
    pre-scaled increment are held to a few ulp here), refinement converges to I0(2) = 2.2795853..., and identical pairs in a
    batch give identical bits."""
    lin = sigkernel_amd.LinearKernel()
    t2 = torch.linspace(0, 1, 2, dtype=torch.float64)[None, :, None].to(DEV).repeat(70, 1, 1).contiguous()
    k0 = sigkernel_amd.SigKernel(lin, 0).compute_kernel(t2, t2)
    assert torch.all(k0 == k0[0]) and abs(float(k0[0]) - 2.25) <= 2e-15
    for M, d in ((2, 3), (9, 2), (128, 1), (128, 3)):
        t = torch.linspace(0, 1, M, dtype=torch.float64)[None, :, None]
        X = t.to(DEV).repeat(70, 1, 1).contiguous()
        k = sigkernel_amd.SigKernel(lin, d).compute_kernel(X, X)
        want = O.gram_forward(t, t, lin, d)[0, 0]
        assert torch.all(k == k[0])
        assert abs(float(k[0]) - want) <= 1e-12 * want
        assert abs(float(k[0]) - 2.2795853023360673) <= 5e-3 / ((M - 1) << d)


def test_long_paths_beyond_the_reference_gpu_limit(be):
    """The reference's GPU path asserts max(MM,NN) < 1024 (sigkernel.py:222,368); ours must not."""
    gen = torch.Generator().manual_seed(1)
    X = walk(gen, 2, 700, 3).to(DEV)
    Y = walk(gen, 2, 640, 3).to(DEV)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
    K = sk.compute_Gram(X, Y)
    want = O.gram_forward(X.cpu(), Y.cpu(), sigkernel_amd.RBFKernel(1.0), 1)
    assert rel_err(K.cpu().numpy(), want) <= 1e-11
