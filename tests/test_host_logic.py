"""Host logic of sigkernel_amd (autograd wiring, tiling, loss formulas) on CPU, with the HIP back-end
replaced by the oracle-backed fake from tests/fake_backend.py.  Expected values: golden fixtures
generated from the real reference."""
import numpy as np
import pytest
import torch

import sigkernel_amd
from conftest import golden, golden_gram_cases, grad_tol, make_kernel, rel_err

FWD_TOL = 1e-13


def _sk(c, **kw):
    return sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"]), _naive_solver=bool(c["naive"]), **kw)


@pytest.mark.parametrize("name", golden_gram_cases())
def test_gram_forward_backward(oracle_backend, name):
    c = golden(name)
    X, Y, w = (torch.from_numpy(c[k]) for k in ("X", "Y", "w"))
    sk = _sk(c)
    assert rel_err(sk.compute_Gram(X, Y).numpy(), c["gram"]) <= FWD_TOL
    Xg = X.clone().requires_grad_(True)
    K = sk.compute_Gram(Xg, Y, sym=False)
    assert K.shape == (X.shape[0], Y.shape[0]) and K.dtype == X.dtype
    (K * w).sum().backward()
    assert rel_err(Xg.grad.numpy(), c["grad_w"]) <= grad_tol(name, "grad_w")


@pytest.mark.parametrize("name", [n for n in golden_gram_cases() if "gram_xx_sym" in golden(n)])
def test_gram_xx_two_times_rule_and_mmd(oracle_backend, name):
    c = golden(name)
    X, Y = torch.from_numpy(c["X"]), torch.from_numpy(c["Y"])
    sk = _sk(c)
    Xg = X.clone().requires_grad_(True)
    G = sk.compute_Gram(Xg, Xg, sym=True)
    assert rel_err(G.detach().numpy(), c["gram_xx_sym"]) <= 1e-12
    G.sum().backward()
    assert rel_err(Xg.grad.numpy(), c["grad_xx_sum"]) <= grad_tol(name, "grad_xx_sum")
    Xg = X.clone().requires_grad_(True)
    mmd = sk.compute_mmd(Xg, Y)
    assert abs(float(mmd.detach()) - float(c["mmd"])) <= 1e-12 * max(1.0, abs(float(c["mmd"])))
    mmd.backward()
    assert rel_err(Xg.grad.numpy(), c["grad_mmd"]) <= grad_tol(name, "grad_mmd")


@pytest.mark.parametrize("name", ["gram_c2mini_rbf_d1", "gram_c3mini_lin_d1"])
def test_symmetric_gram_triangular_blocks_with_gradient(oracle_backend, name, monkeypatch):
    """compute_Gram(X, X, sym=True) with a gradient in row blocks (only the pairs on and above the diagonal are solved; the
    mirror pairs' share comes from the second-argument contraction): same values and gradients as the reference's fixtures,
    also for a non-symmetric upstream gradient against the full sym=False computation."""
    from sigkernel_amd import sigkernel as S
    monkeypatch.setattr(S, "_SYM_TILES", 3)
    monkeypatch.setattr(S, "_SYM_MIN_CELLS", 0.0)
    monkeypatch.setattr(S, "_SYM_MIN_ROWS", 1)
    monkeypatch.setattr(S, "_gram_symmetric", S._gram_symmetric)
    c = golden(name)
    X = torch.from_numpy(c["X"])
    # force the block route even for 5..6 paths: the A >= 8 T rule is for launch overheads, not correctness
    orig = S._gram_symmetric

    def blocks(be, static_kernel, Xd, dyadic_order, naive, workspace_bytes, keep_blocks=None):
        monkeypatch.setattr(S, "_SYM_TILES", 3)
        A = Xd.shape[0]
        if keep_blocks is None:
            return orig(be, static_kernel, Xd, dyadic_order, naive, workspace_bytes, keep_blocks)
        K = torch.empty(A, A, dtype=Xd.dtype)
        step = -(-A // 3)
        for r0 in range(0, A, step):
            r1 = min(r0 + step, A)
            kept = []
            blk = S._gram_block(be, static_kernel, Xd[r0:r1].contiguous(), Xd[r0:].contiguous(), dyadic_order, naive, workspace_bytes, 3, kept)
            keep_blocks.append((r0, r1, kept))
            K[r0:r1, r0:] = blk
            if r1 < A:
                K[r1:, r0:r1] = blk[:, r1 - r0:].t()
        iu = torch.triu_indices(A, A, offset=1)
        K[iu[1], iu[0]] = K[iu[0], iu[1]]
        return K
    monkeypatch.setattr(S, "_gram_symmetric", blocks)
    sk = _sk(c)
    Xg = X.clone().requires_grad_(True)
    G = sk.compute_Gram(Xg, Xg, sym=True)
    assert rel_err(G.detach().numpy(), c["gram_xx_sym"]) <= 1e-12
    G.sum().backward()
    assert rel_err(Xg.grad.numpy(), c["grad_xx_sum"]) <= grad_tol(name, "grad_xx_sum")
    w = torch.randn(X.shape[0], X.shape[0], generator=torch.Generator().manual_seed(1), dtype=torch.float64)   # not symmetric
    X1 = X.clone().requires_grad_(True)
    (sk.compute_Gram(X1, X1, sym=True) * w).sum().backward()
    monkeypatch.setattr(S, "_gram_symmetric", orig)
    X2 = X.clone().requires_grad_(True)
    (sk.compute_Gram(X2, X2, sym=False) * w).sum().backward()
    assert rel_err(X1.grad.numpy(), X2.grad.numpy()) <= 1e-12


@pytest.mark.parametrize("name", golden_gram_cases())
def test_paired_kernel(oracle_backend, name):
    c = golden(name)
    n = c["paired"].shape[0]
    X, Y, wp = torch.from_numpy(c["X"][:n]), torch.from_numpy(c["Y"][:n]), torch.from_numpy(c["wp"])
    sk = _sk(c)
    Xg = X.clone().requires_grad_(True)
    K = sk.compute_kernel(Xg, Y)
    assert K.shape == (n,)
    assert rel_err(K.detach().numpy(), c["paired"]) <= FWD_TOL
    (K * wp).sum().backward()
    assert rel_err(Xg.grad.numpy(), c["grad_paired"]) <= grad_tol(name, "grad_paired")


def test_readme_example(oracle_backend):
    c = golden("readme_c1")
    X, Y, Z = (torch.from_numpy(c[k]) for k in ("X", "Y", "Z"))
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(sigma=float(c["sigma"])), dyadic_order=int(c["dyadic"]))
    assert rel_err(sk.compute_kernel(X, Y).numpy(), c["kernel"]) <= FWD_TOL
    assert rel_err(sk.compute_Gram(X, Y, sym=False).numpy(), c["gram"]) <= FWD_TOL
    Xg = X.clone().requires_grad_(True)
    sk.compute_kernel(Xg, Y).sum().backward()
    assert rel_err(Xg.grad.numpy(), c["grad_kernel_sum"]) <= grad_tol("readme_c1", "grad_kernel_sum")
    Xg = X.clone().requires_grad_(True)
    mmd = sk.compute_mmd(Xg, Y)
    mmd.backward()
    assert abs(float(mmd.detach()) - float(c["mmd"])) <= 1e-13
    assert rel_err(Xg.grad.numpy(), c["grad_mmd"]) <= grad_tol("readme_c1", "grad_mmd")
    assert abs(float(sk.compute_distance(X, Y)) - float(c["distance"])) <= 1e-13
    assert abs(float(sk.compute_scoring_rule(X, Z[:1])) - float(c["scoring_rule"])) <= 1e-13
    assert abs(float(sk.compute_expected_scoring_rule(X, Z)) - float(c["expected_scoring_rule"])) <= 1e-13


@pytest.mark.parametrize("name", ["gram_c2mini_rbf_d1", "gram_c3mini_lin_d1", "gram_c4mini_rbf_d2"])
def test_merged_loss_route_equals_the_reference_composition(oracle_backend, name, monkeypatch):
    """compute_mmd / compute_scoring_rule / compute_expected_scoring_rule of training-sized batches go through ONE Gram block
    K(X, [X; Y]) (sigkernel._SigKernelLoss) instead of the reference's two or three compute_Gram calls (sigkernel.py:146-197):
    same values (summation order aside) and gradients as that composition (routes.no_merged_loss) and as the reference's fixtures;
    paths of different lengths, single paths and big batches keep the composition."""
    from sigkernel_amd import sigkernel as S
    c = golden(name)
    n = min(c["X"].shape[1], c["Y"].shape[1])
    X, Y = torch.from_numpy(c["X"][:, :n].copy()), torch.from_numpy(c["Y"][:, :n].copy())
    sk = _sk(c)
    blocks = []
    real = S._gram_block
    monkeypatch.setattr(S, "_gram_block", lambda be, k, Xd, Yd, *a, **kw: (blocks.append((Xd.shape[0], Yd.shape[0])), real(be, k, Xd, Yd, *a, **kw))[1])
    for fn, Yv, yy in ((sk.compute_mmd, Y, True), (sk.compute_expected_scoring_rule, Y, False), (sk.compute_scoring_rule, Y[:1], False)):
        A, B = X.shape[0], Yv.shape[0]
        got, grads = [], []
        for composed in (False, True):
            monkeypatch.setattr(sigkernel_amd.routes, "no_merged_loss", composed)
            del blocks[:]
            Xg = X.clone().requires_grad_(True)
            v = fn(Xg, Yv)
            v.backward()
            got.append(float(v.detach())), grads.append(Xg.grad.numpy().copy())
            if not composed:
                assert blocks == ([(B, B)] if yy else []) + [(A, A + B)], blocks          # one block for K_XX and K_XY (+ K_YY's)
            with torch.no_grad():
                assert abs(float(fn(X, Yv)) - got[-1]) <= 1e-13 * max(1.0, abs(got[-1]))
        assert abs(got[0] - got[1]) <= 1e-13 * max(1.0, abs(got[1]))
        assert rel_err(grads[0], grads[1]) <= 1e-12
        if fn == sk.compute_mmd and n == c["X"].shape[1] == c["Y"].shape[1]:
            assert abs(got[0] - float(c["mmd"])) <= 1e-12 * max(1.0, abs(float(c["mmd"])))
            assert rel_err(grads[0], c["grad_mmd"]) <= grad_tol(name, "grad_mmd")
    # outside the merged route: different lengths, a process group's business, too many pairs, one path
    monkeypatch.setattr(sigkernel_amd.routes, "no_merged_loss", False)
    assert sk._merged_loss(X, Y[:, :n - 1], True) is None
    assert sk._merged_loss(X[:1], Y, True) is None and sk._merged_loss(X, Y[:1], True) is None and sk._merged_loss(X, Y[:1], False) is not None
    monkeypatch.setattr(S, "_SYM_MIN_CELLS", float(X.shape[0]) ** 2 * ((n - 1) << int(c["dyadic"])) ** 2)
    assert sk._merged_loss(X, Y, True) is None


def test_compute_distance_merged_pairs_equal_the_reference_composition(oracle_backend, monkeypatch):
    """compute_distance of a training-sized batch solves k(x_i, x_i) and k(x_i, y_i) as one paired batch of 2n pairs: the same value
    and gradient as the reference's three compute_kernel calls (sigkernel.py:130-144; routes.no_merged_loss)."""
    c = golden("gram_c3mini_lin_d1")
    n = min(c["X"].shape[0], c["Y"].shape[0])
    m = min(c["X"].shape[1], c["Y"].shape[1])
    X, Y = torch.from_numpy(c["X"][:n, :m].copy()), torch.from_numpy(c["Y"][:n, :m].copy())
    out = []
    for composed in (False, True):
        monkeypatch.setattr(sigkernel_amd.routes, "no_merged_loss", composed)
        Xg = X.clone().requires_grad_(True)
        v = _sk(c).compute_distance(Xg, Y)
        v.backward()
        out.append((float(v.detach()), Xg.grad.numpy().copy()))
    assert abs(out[0][0] - out[1][0]) <= 1e-14 * max(1.0, abs(out[1][0]))
    assert rel_err(out[0][1], out[1][1]) <= 1e-13


def test_results_do_not_depend_on_tiling_or_max_batch(oracle_backend):
    c = golden("gram_c3mini_lin_d1")
    X, Y, w = (torch.from_numpy(c[k]) for k in ("X", "Y", "w"))
    ref = _sk(c).compute_Gram(X, Y)
    tiny = _sk(c, workspace_bytes=1)          # one Gram row per tile
    assert torch.equal(tiny.compute_Gram(X, Y, max_batch=2), ref)
    Xa = X.clone().requires_grad_(True)
    Xb = X.clone().requires_grad_(True)
    (_sk(c).compute_Gram(Xa, Y) * w).sum().backward()
    (tiny.compute_Gram(Xb, Y, max_batch=1) * w).sum().backward()
    assert rel_err(Xb.grad.numpy(), Xa.grad.numpy()) <= 1e-14
    assert torch.equal(tiny.compute_kernel(X[:4], Y[:4]), _sk(c).compute_kernel(X[:4], Y[:4]))


@pytest.mark.parametrize("name", ["gram_c3mini_lin_d1", "gram_c2mini_rbf_d1"])
def test_more_pairs_than_one_launch_indexes_are_tiled_over_rows(oracle_backend, name, monkeypatch):
    """The fused kernels index pairs with 32 bits; a Gram call beyond _MAX_LAUNCH_PAIRS is solved in row tiles (forward, kept edges and
    backward alike), with the same values and gradients."""
    from sigkernel_amd import sigkernel as S
    c = golden(name)
    X, Y, w = (torch.from_numpy(c[k]) for k in ("X", "Y", "w"))
    want = _sk(c).compute_Gram(X, Y)
    Xa = X.clone().requires_grad_(True)
    (_sk(c).compute_Gram(Xa, Y) * w).sum().backward()
    blocks = []
    real = S._gram_block
    monkeypatch.setattr(S, "_MAX_LAUNCH_PAIRS", 2 * Y.shape[0] + 1)            # two rows per tile
    monkeypatch.setattr(S, "_gram_block", lambda be, k, Xd, Yd, *a, **kw: (blocks.append(Xd.shape[0]), real(be, k, Xd, Yd, *a, **kw))[1])
    assert torch.equal(_sk(c).compute_Gram(X, Y), want)
    assert blocks[0] == X.shape[0] and all(b <= 2 for b in blocks[1:]) and sum(blocks[1:]) == X.shape[0]
    Xb = X.clone().requires_grad_(True)
    (_sk(c).compute_Gram(Xb, Y) * w).sum().backward()
    assert rel_err(Xb.grad.numpy(), Xa.grad.numpy()) <= 1e-13
    assert rel_err(Xb.grad.numpy(), c["grad_w"]) <= grad_tol(name, "grad_w")
    assert S._cap_rows(10, 1000, 10 ** 6) == 10 ** 9 // (2 * Y.shape[0] + 1) + 1      # a budget tile never exceeds the pair limit


def test_no_gradient_for_second_argument_and_asserts(oracle_backend):
    c = golden("gram_lin_d0_ragged")
    X, Y = torch.from_numpy(c["X"]), torch.from_numpy(c["Y"])
    sk = _sk(c)
    Xg, Yg = X.clone().requires_grad_(True), Y.clone().requires_grad_(True)
    sk.compute_Gram(Xg, Yg).sum().backward()
    assert Yg.grad is None                      # sigkernel.py:412,416 return None for Y
    with pytest.raises(AssertionError, match="second input should not require grad"):
        sk.compute_mmd(X, Yg)
    with pytest.raises(AssertionError):
        sk.compute_distance(X[:3], Yg[:3])


def test_single_point_paths(oracle_backend):
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1)
    X = torch.rand(3, 1, 2, dtype=torch.float64, requires_grad=True)
    Y = torch.rand(4, 5, 2, dtype=torch.float64)
    K = sk.compute_Gram(X, Y)
    assert torch.equal(K, torch.ones(3, 4, dtype=torch.float64))
    K.sum().backward()
    assert torch.equal(X.grad, torch.zeros_like(X))


def test_input_validation(oracle_backend):
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 0)
    with pytest.raises(ValueError):
        sk.compute_Gram(torch.rand(2, 3, 2), torch.rand(2, 3, 3))
    with pytest.raises(ValueError):
        sk.compute_kernel(torch.rand(2, 3, 2), torch.rand(3, 3, 2))
    with pytest.raises(ValueError):
        sk.compute_Gram(torch.rand(2, 3), torch.rand(2, 3, 3))


def _lib_route(*args):
    from sigkernel_amd import _lib
    return _lib.HipBackend.route(*args)


class _FusedFake:
    """The oracle-backed fake with a fused linear adjoint that declines the case (returns None, as the HIP one does outside its
    scope): exercises the re-tiled fallback of sigkernel._rows_gradient on CPU.  (Exploding kernels are no longer a reason to
    fall back: the library rescues such pairs on the device, sk_adj_fused_rescue.hip.)"""

    def __init__(self, base, residual):
        self._base, self._residual, self.fused_calls, self.tile_rows = base, residual, 0, []
        self.ADJ_RESIDUAL_TOL = 1e-8

    def __getattr__(self, name):
        return getattr(self._base, name)

    def solve_fwd_fused_linear(self, X, Y, scale, dyadic, naive, gram, keep_edges=False):
        return None          # forward takes the tiled route; the fused adjoint asks for edges itself

    route = staticmethod(_lib_route)     # the library's own answer (sk_route_query is host code: no GPU needed)

    def linear_adjoint_fused(self, X, Y, param, dyadic, edges, scale, gram=True, kfinal=None, naive=False):
        self.fused_calls += 1
        self.got_kfinal = kfinal is not None
        return None

    def static_increments(self, kind, param, X, Y, gram):
        self.tile_rows.append(X.shape[0])
        return self._base.static_increments(kind, param, X, Y, gram)


def test_failed_fused_adjoint_falls_back_tiled_by_the_unfused_budget():
    """when the fused linear adjoint does not cover the case the backward pass must take the unfused route in tiles
    sized for THAT route's transient memory; and the fused adjoint must be handed the forward values that arm its rescue."""
    from sigkernel_amd import _lib, sigkernel as skmod
    from fake_backend import OracleBackend
    c = golden("gram_c3mini_lin_d1")
    X, Y, w = (torch.from_numpy(c[k]) for k in ("X", "Y", "w"))
    A, M, B, N = X.shape[0], X.shape[1], Y.shape[0], Y.shape[1]
    budget = 3 * B * M * N * 8 * 2          # room for two rows of the unfused route
    fake = _FusedFake(OracleBackend(), residual=1.0)
    # the fused route needs a forward that keeps edges: hand the fake one that returns a token
    fake.solve_fwd_fused_linear = lambda X, Y, scale, dyadic, naive, gram, keep_edges=False: (
        (torch.zeros(X.shape[0], Y.shape[0], dtype=X.dtype), torch.zeros(1)) if keep_edges else None)
    prev = _lib.set_backend(fake)
    try:
        sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), int(c["dyadic"]), workspace_bytes=budget)
        Xg = X.clone().requires_grad_(True)
        K = sk.compute_Gram(Xg, Y)
        fake.tile_rows.clear()
        (K * w).sum().backward()
    finally:
        _lib.set_backend(prev)
    assert fake.fused_calls >= 1 and fake.got_kfinal               # tried (in its own, larger tiles), declined -> fallback
    assert fake.tile_rows and max(fake.tile_rows) <= 2 and sum(fake.tile_rows) == A, fake.tile_rows
    assert rel_err(Xg.grad.numpy(), c["grad_w"]) <= grad_tol("gram_c3mini_lin_d1", "grad_w")


@pytest.mark.parametrize("kind", ["linear", "rbf"])
def test_streaming_backward_reuses_the_increments_the_forward_kept(kind, monkeypatch):
    """sigkernel._gram_block / _tile_gradient: a one-tile Gram block on the streaming route with a gradient pending keeps the increments its
    fused static kernel formed, beside the edges, when they fit keep_increments_fraction of the budget -- backward then does not form them
    again; several tiles, a fraction of 0, or increments the torch route formed (no kernel for them) and they are formed again.  Same
    gradient every way, equal to the golden one."""
    from sigkernel_amd import _lib, sigkernel as skmod
    from fake_backend import OracleBackend
    name = "gram_c3mini_lin_d1" if kind == "linear" else "gram_c2mini_rbf_d1"
    c = golden(name)
    X, Y, w = (torch.from_numpy(c[k]) for k in ("X", "Y", "w"))
    A, M, B, N = X.shape[0], X.shape[1], Y.shape[0], Y.shape[1]
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(float(c["param"]))

    calls = []

    class Counting(OracleBackend):
        def static_increments(self, *a, **kw):
            calls.append(1)
            return OracleBackend.static_increments(self, *a, **kw)

    def step(workspace=None, be=None):
        del calls[:]
        prev = _lib.set_backend(be or Counting())
        try:
            Xg = X.clone().requires_grad_(True)
            (sigkernel_amd.SigKernel(k, int(c["dyadic"]), workspace_bytes=workspace).compute_Gram(Xg, Y) * w).sum().backward()
        finally:
            _lib.set_backend(prev)
        return Xg.grad, len(calls)
    g_kept, n_kept = step()
    assert n_kept == 1 and rel_err(g_kept.numpy(), c["grad_w"]) <= grad_tol(name, "grad_w")
    monkeypatch.setattr(skmod, "_KEEP_INCREMENTS_FRACTION", 0.0)
    g0, n0 = step()
    assert n0 == 2 and torch.equal(g0, g_kept)
    monkeypatch.setattr(skmod, "_KEEP_INCREMENTS_FRACTION", None)
    g_tiles, n_tiles = step(workspace=3 * B * M * N * 8 * 2)      # two rows per tile: nothing kept across tiles
    assert n_tiles >= 2 * ((A + 1) // 2) and torch.allclose(g_tiles, g_kept, rtol=1e-12, atol=1e-300)

    # the symmetric Gram's row blocks (triangle + second-argument sums) reuse theirs the same way
    monkeypatch.setattr(skmod, "_SYM_TILES", 2)
    monkeypatch.setattr(skmod, "_SYM_MIN_CELLS", 0.0)
    monkeypatch.setattr(skmod, "_SYM_MIN_ROWS", 1)
    gen = torch.Generator().manual_seed(5)
    X16 = torch.cat([X, X + 0.01 * torch.randn(X.shape, generator=gen, dtype=X.dtype), X.flip(1), 0.5 * X])[:16].contiguous()

    def sym_step():
        del calls[:]
        prev = _lib.set_backend(Counting())
        try:
            Xg = X16.clone().requires_grad_(True)
            sigkernel_amd.SigKernel(k, int(c["dyadic"])).compute_Gram(Xg, Xg, sym=True).sum().backward()
        finally:
            _lib.set_backend(prev)
        return Xg.grad, len(calls)
    gs_kept, ns_kept = sym_step()
    monkeypatch.setattr(skmod, "_KEEP_INCREMENTS_FRACTION", 0.0)
    gs0, ns0 = sym_step()
    monkeypatch.setattr(skmod, "_KEEP_INCREMENTS_FRACTION", None)
    assert ns_kept == 2 and ns0 == 4 and torch.equal(gs_kept, gs0), (ns_kept, ns0)

    class NoKernel(Counting):      # (a path dimension the static kernels do not cover: increments by the torch route, never kept)
        def static_increments(self, *a, **kw):
            calls.append(1)
            return None
    g_nk, n_nk = step(be=NoKernel())
    assert n_nk >= 2 and torch.allclose(g_nk, g_kept, rtol=1e-12, atol=1e-300)
