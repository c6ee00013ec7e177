"""Directional-derivative path: k, k_gamma, k_gamma_gamma.

Golden vectors: tests/golden/derivatives.npz, produced by the reference's own MPS solver run on CPU tensors
(make_golden.py section 6; the reference's CPU dispatch for this function is broken, the CUDA/MPS stencil
defines the behaviour).

Tolerances.  The solver itself is compared at 1e-11 (fast kernels) / bit-exact (SK_FLAG_EXACT).  End to end,
the reference differentiates the static kernel by finite differences with eps = 1e-4 and scales the
differences by 1/eps and 1/eps^2 = 1e8: a last-bit difference in a static Gram entry (torch CPU vs rocBLAS,
summation order) becomes ~1e-8 in an increment of the second derivative and ~1e-12 in the first.  Hence
1e-9 for k_gamma and 2e-6 for k_gamma_gamma against the fixtures, stated here and in DESIGN.md.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden, rel_err, walk

TOL_K, TOL_KD, TOL_KDD = 1e-12, 1e-9, 2e-6


def _kernel(c, name):
    import sigkernel_amd
    return (sigkernel_amd.LinearKernel() if str(c[name + "_kernel"]) == "linear"
            else sigkernel_amd.RBFKernel(float(c[name + "_param"])))


def _api_cases():
    return [str(n) for n in golden("derivatives")["api_cases"]]


# ------------------------------------------------------------------------------------------------ oracle (CPU)
def test_oracle_derivative_grids_bit_identical_to_reference():
    from oracle import oracle as O
    c = golden("derivatives")
    k, kd, kdd, grids = O.solve_deriv_coarse(c["inc"], c["inc_d"], c["inc_dd"], 0, want_grid=True)
    for i, n in enumerate(("K", "Kd", "Kdd")):
        assert np.array_equal(grids[i], c[n]), n
    assert np.array_equal(k, c["K"][..., -1, -1]) and np.array_equal(kdd, c["Kdd"][..., -1, -1])


def test_oracle_dyadic_refinement_by_index_equals_tiled_increments():
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    a = [rng.normal(scale=0.4, size=(3, 5, 4)) for _ in range(3)]
    for d in (1, 2):
        r = 1 << d
        fine = [np.repeat(np.repeat(x, r, axis=-2), r, axis=-1) / r / r for x in a]
        got = O.solve_deriv_coarse(a[0], a[1], a[2], d)
        ref = O.solve_deriv_coarse(fine[0], fine[1], fine[2], 0)
        for g, e in zip(got, ref):
            assert np.array_equal(g, e)


@pytest.mark.parametrize("name", _api_cases())
def test_oracle_kgrad_matches_reference(name):
    from oracle import oracle as O
    c = golden("derivatives")
    X, Y, g = (torch.from_numpy(c[name + s]) for s in ("_X", "_Y", "_gamma"))
    k, kd, kdd = O.kgrad(X, Y, g, _kernel(c, name), int(c[name + "_dyadic"]))
    assert rel_err(k, c[name + "_k"]) <= TOL_K
    assert rel_err(kd, c[name + "_kd"]) <= TOL_KD
    assert rel_err(kdd, c[name + "_kdd"]) <= TOL_KDD


def test_oracle_derivatives_are_derivatives():
    """k_gamma and k_gamma_gamma converge, as the grid is refined, to central differences of k along gamma (the coupled
    stencil is not the derivative of the discrete K scheme: 17 % / 1.5 % / 0.1 % apart at dyadic order 0 / 1 / 2; the
    floor is the O(eps) truncation of the reference's one-sided static-kernel differences)."""
    from oracle import oracle as O
    import sigkernel_amd
    gen = torch.Generator().manual_seed(2)
    X, Y, g = walk(gen, 3, 8, 2) * 2, walk(gen, 2, 7, 2) * 2, torch.randn(3, 8, 2, generator=gen, dtype=torch.float64)
    sk = sigkernel_amd.RBFKernel(1.0)
    k, kd, kdd = O.kgrad(X, Y, g, sk, 3)
    h = 1e-3
    kp, km = O.gram_forward(X + h * g, Y, sk, 3), O.gram_forward(X - h * g, Y, sk, 3)
    assert rel_err(kd, (kp - km) / (2 * h)) <= 2e-3
    assert rel_err(kdd, (kp - 2 * k + km) / h ** 2) <= 2e-3


# ------------------------------------------------------------------------------------------------ host logic (CPU, fake back-end)
@pytest.mark.parametrize("name", _api_cases())
def test_api_on_fake_backend(oracle_backend, name):
    import sigkernel_amd
    c = golden("derivatives")
    X, Y, g = (torch.from_numpy(c[name + s]) for s in ("_X", "_Y", "_gamma"))
    sk = sigkernel_amd.SigKernel(_kernel(c, name), int(c[name + "_dyadic"]))
    k, kd, kdd = sk.compute_kernel_and_derivatives_Gram(X, Y, g)
    assert k.shape == kd.shape == kdd.shape == (X.shape[0], Y.shape[0])
    assert rel_err(k, c[name + "_k"]) <= TOL_K and rel_err(kd, c[name + "_kd"]) <= TOL_KD
    assert rel_err(kdd, c[name + "_kdd"]) <= TOL_KDD
    # tiling by a tiny HBM budget does not change anything
    sk2 = sigkernel_amd.SigKernel(_kernel(c, name), int(c[name + "_dyadic"]), workspace_bytes=1)
    for a, b in zip(sk2.compute_kernel_and_derivatives_Gram(X, Y, g), (k, kd, kdd)):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        sk.compute_kernel_and_derivatives_Gram(X, Y, g[:, :-1])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sigkernel_amd
        from sigkernel_amd import _lib
        from fake_backend import OracleBackend
        _lib.set_backend(OracleBackend())
        c = golden("derivatives")
        X, Y, g = (torch.from_numpy(c[name + s]) for s in ("_X", "_Y", "_gamma"))
        sk = sigkernel_amd.SigKernel(_kernel(c, name), int(c[name + "_dyadic"]), process_group=dist.group.WORLD)
        k, kd, kdd = sk.compute_kernel_and_derivatives_Gram(X, Y, g)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), k=k.numpy(), kd=kd.numpy(), kdd=kdd.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("lin_d1", 2), ("rbf_d2", 4)])
def test_sharded_derivatives_over_gloo(tmp_path, name, world):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    c = golden("derivatives")
    for r in range(world):
        got = dict(np.load(tmp_path / ("rank%d.npz" % r)))
        assert rel_err(got["k"], c[name + "_k"]) <= TOL_K
        assert rel_err(got["kd"], c[name + "_kd"]) <= TOL_KD
        assert rel_err(got["kdd"], c[name + "_kdd"]) <= TOL_KDD


# ------------------------------------------------------------------------------------------------ GPU parity (through the C ABI)
def _oracle_solve(inc3, d):
    from oracle import oracle as O
    a = inc3.detach().double().cpu().numpy()
    return O.solve_deriv_coarse(a[0], a[1], a[2], d, nthreads=8)


DERIV_SHAPES = [  # P, Mc, Nc, dyadic
    (3, 1, 1, 0), (5, 7, 9, 0), (4, 9, 7, 1), (6, 5, 11, 2), (2, 3, 4, 3),
    (9, 16, 16, 1), (3, 63, 63, 1), (2, 64, 33, 0), (2, 65, 70, 1), (2, 127, 127, 1), (1, 130, 40, 2),
    (2, 40, 300, 0), (70, 8, 8, 1), (1, 200, 17, 1),
]


def _rand_inc3(P, Mc, Nc, seed, dtype=torch.float64):
    gen = torch.Generator().manual_seed(seed)
    s = 1.5 / np.sqrt(Mc * Nc)
    return (torch.randn(3, P, Mc, Nc, generator=gen, dtype=torch.float64) * s).to(dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("P,Mc,Nc,d", DERIV_SHAPES)
def test_gpu_deriv_solver_exact_flag_is_bit_identical_to_oracle(P, Mc, Nc, d):
    from sigkernel_amd import _lib
    inc3 = _rand_inc3(P, Mc, Nc, 10 + Mc)
    got = _lib.get_backend().solve_deriv(inc3.cuda(), d, flags=_lib.FLAG_EXACT)
    for g, e in zip(got, _oracle_solve(inc3, d)):
        assert np.array_equal(g.cpu().numpy(), e)


@pytest.mark.gpu
@pytest.mark.parametrize("P,Mc,Nc,d", DERIV_SHAPES)
def test_gpu_deriv_solver_default_path(P, Mc, Nc, d):
    from sigkernel_amd import _lib
    inc3 = _rand_inc3(P, Mc, Nc, 20 + Nc)
    be = _lib.get_backend()
    dev = inc3.cuda()
    # rows padded to whole 128-byte lines (what deriv_increments produces) and dense rows (generic callers)
    ld = (Nc + 15) // 16 * 16
    padded = torch.zeros(3, P, Mc, ld, dtype=torch.float64, device="cuda")
    padded[..., :Nc] = dev
    ref = _oracle_solve(inc3, d)
    for src in (padded[..., :Nc], dev):
        for g, e in zip(be.solve_deriv(src, d), ref):
            assert rel_err(g.cpu().numpy(), e) <= 1e-11


@pytest.mark.gpu
def test_gpu_deriv_fuzz_random_shapes():
    """40 random shapes (ragged, tiny, multi-band, partial lane groups) through the fast kernel, fp64 and fp32."""
    from sigkernel_amd import _lib
    be = _lib.get_backend()
    rng = np.random.default_rng(99)
    for it in range(40):
        d = int(rng.integers(0, 3))
        Mc, Nc, P = int(rng.integers(1, 300 >> d)), int(rng.integers(1, 300 >> d)), int(rng.integers(1, 30))
        inc3 = _rand_inc3(P, Mc, Nc, 1000 + it)
        ld = (Nc + 15) // 16 * 16
        padded = torch.zeros(3, P, Mc, ld, dtype=torch.float64, device="cuda")
        padded[..., :Nc] = inc3.cuda()
        ref = _oracle_solve(inc3, d)
        for g, e in zip(be.solve_deriv(padded[..., :Nc], d, flags=_lib.FLAG_FAST_ONLY), ref):
            assert rel_err(g.cpu().numpy(), e) <= 1e-11, (it, P, Mc, Nc, d)
        if d <= 1:
            p32 = torch.zeros(3, P, Mc, (Nc + 31) // 32 * 32, dtype=torch.float32, device="cuda")
            p32[..., :Nc] = inc3.float().cuda()
            ref32 = _oracle_solve(inc3.float(), d)
            for g, e in zip(be.solve_deriv(p32[..., :Nc], d, flags=_lib.FLAG_FAST_ONLY), ref32):
                assert rel_err(g.cpu().numpy(), e) <= 1e-5, (it, P, Mc, Nc, d)


@pytest.mark.gpu
def test_gpu_deriv_fast_kernel_is_the_one_that_runs():
    """SK_FLAG_FAST_ONLY must succeed on line-padded inputs in the fast kernel's scope (no silent fallback)."""
    from sigkernel_amd import _lib
    be = _lib.get_backend()
    for (P, Mc, Nc, d) in [(300, 63, 63, 1), (64, 127, 127, 1), (50, 31, 40, 0), (20, 20, 20, 2)]:
        inc3 = _rand_inc3(P, Mc, Nc, 5)
        ld = (Nc + 15) // 16 * 16
        padded = torch.zeros(3, P, Mc, ld, dtype=torch.float64, device="cuda")
        padded[..., :Nc] = inc3.cuda()
        got = be.solve_deriv(padded[..., :Nc], d, flags=_lib.FLAG_FAST_ONLY)
        for g, e in zip(got, _oracle_solve(inc3, d)):
            assert rel_err(g.cpu().numpy(), e) <= 1e-11


@pytest.mark.gpu
def test_gpu_deriv_fp32():
    from sigkernel_amd import _lib
    be = _lib.get_backend()
    for (P, Mc, Nc, d) in [(40, 31, 31, 1), (10, 60, 70, 0), (6, 20, 24, 2)]:
        inc3 = _rand_inc3(P, Mc, Nc, 8, torch.float32)
        ref = _oracle_solve(inc3, d)          # fp64 oracle on the up-cast inputs
        for g, e in zip(be.solve_deriv(inc3.cuda(), d), ref):
            assert g.dtype == torch.float32
            assert rel_err(g.cpu().numpy(), e) <= 1e-5


@pytest.mark.gpu
def test_gpu_deriv_increments_match_reference_order():
    from sigkernel_amd import _lib
    from fake_backend import OracleBackend
    gen = torch.Generator().manual_seed(4)
    G = [torch.randn(5, 9, 12, generator=gen, dtype=torch.float64) for _ in range(3)]
    G[1] = G[0] + 1e-4 * G[1]
    G[2] = G[0] + 2e-4 * G[2]
    got = _lib.get_backend().deriv_increments(*(g.cuda() for g in G), 1e-4)
    ref = OracleBackend().deriv_increments(*G, 1e-4)
    assert got.shape == ref.shape and got.stride(-2) % 16 == 0
    assert torch.equal(got.cpu(), ref)       # same operand order, no FMA contraction: bit-identical


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,M,N", [(0, 3, 9, 12), (1, 2, 20, 70), (1, 8, 33, 64), (0, 32, 5, 130)])
def test_gpu_fused_static_deriv_increments_against_generic_route(kind, D, M, N):
    """sk_static_deriv_increments_* (static kernel fused in) against Gram_matrix x 3 + sk_deriv_increments_*.  The two
    evaluate the static kernel in different summation orders, and the 1/eps, 1/eps^2 scaling amplifies that last-bit
    difference of G: tolerances 1e-14 / 1e-10 / 1e-6 times max|G| (absolute -- for the linear kernel the second-derivative
    increments are round-off around an exact zero)."""
    import sigkernel_amd
    from sigkernel_amd import _lib
    be = _lib.get_backend()
    gen = torch.Generator().manual_seed(M)
    X, Y = (walk(gen, 4, M, D) * 2).cuda(), (walk(gen, 3, N, D) * 2).cuda()
    g = torch.randn(4, M, D, generator=gen, dtype=torch.float64).cuda()
    eps = 1e-4
    k = sigkernel_amd.LinearKernel() if kind == 0 else sigkernel_amd.RBFKernel(0.7)
    fused = be.static_deriv_increments(kind, 1.0 if kind == 0 else 0.7, X, X + eps * g, X + 2. * eps * g, Y, eps)
    ref = be.deriv_increments(k.Gram_matrix(X, Y).contiguous(), k.Gram_matrix(X + eps * g, Y).contiguous(),
                              k.Gram_matrix(X + 2. * eps * g, Y).contiguous(), eps)
    assert fused.shape == ref.shape == (3, 4, 3, M - 1, N - 1) and fused.stride(-2) % 16 == 0
    gmax = float(k.Gram_matrix(X, Y).abs().max())
    for i, tol in enumerate((1e-14, 1e-10, 1e-6)):
        assert float((fused[i] - ref[i]).abs().max()) <= tol * max(gmax, 1.0), i
    # the padding columns the solver streams are zero
    ld = fused.stride(-2)
    raw = torch.as_strided(fused, (3, 4, 3, M - 1, ld), fused.stride())
    assert float(raw[..., N - 1:].abs().max()) == 0.0 if ld > N - 1 else True


@pytest.mark.gpu
def test_gpu_user_defined_static_kernel_takes_the_generic_route():
    import sigkernel_amd

    class MyRBF(sigkernel_amd.RBFKernel):     # a subclass is never fused
        pass
    gen = torch.Generator().manual_seed(1)
    X, Y = (walk(gen, 5, 12, 3) * 2).cuda(), (walk(gen, 4, 10, 3) * 2).cuda()
    g = torch.randn(5, 12, 3, generator=gen, dtype=torch.float64).cuda()
    a = sigkernel_amd.SigKernel(MyRBF(0.8), 1).compute_kernel_and_derivatives_Gram(X, Y, g)
    b = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.8), 1).compute_kernel_and_derivatives_Gram(X, Y, g)
    for u, v, tol in zip(a, b, (TOL_K, TOL_KD, TOL_KDD)):
        assert rel_err(u.cpu(), v.cpu()) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("name", _api_cases())
def test_gpu_api_matches_reference_fixture(name):
    import sigkernel_amd
    c = golden("derivatives")
    X, Y, g = (torch.from_numpy(c[name + s]).cuda() for s in ("_X", "_Y", "_gamma"))
    sk = sigkernel_amd.SigKernel(_kernel(c, name), int(c[name + "_dyadic"]))
    k, kd, kdd = sk.compute_kernel_and_derivatives_Gram(X, Y, g)
    assert k.device.type == "cuda" and k.shape == (X.shape[0], Y.shape[0])
    assert rel_err(k.cpu(), c[name + "_k"]) <= TOL_K
    assert rel_err(kd.cpu(), c[name + "_kd"]) <= TOL_KD
    assert rel_err(kdd.cpu(), c[name + "_kdd"]) <= TOL_KDD
    k0 = sk.compute_Gram(X, Y)
    assert rel_err(k.cpu(), k0.cpu()) <= 1e-12   # the K state is the plain signature kernel


@pytest.mark.gpu
def test_gpu_api_larger_problem_against_oracle_and_linearity():
    """128 x 96 pairs of length 64: against the CPU oracle, plus properties that hold at any size -- k_gamma is linear
    and k_gamma_gamma quadratic in gamma (exactly so for the linear static kernel up to FD round-off)."""
    import sigkernel_amd
    from oracle import oracle as O
    gen = torch.Generator().manual_seed(9)
    X, Y = walk(gen, 128, 64, 4), walk(gen, 96, 64, 4)
    g = torch.randn(128, 64, 4, generator=gen, dtype=torch.float64) / 8
    lin = sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(lin, 1)
    k, kd, kdd = (t.cpu().numpy() for t in sk.compute_kernel_and_derivatives_Gram(X.cuda(), Y.cuda(), g.cuda()))
    ek, ekd, ekdd = O.kgrad(X, Y, g, lin, 1, nthreads=8)
    assert rel_err(k, ek) <= TOL_K and rel_err(kd, ekd) <= TOL_KD and rel_err(kdd, ekdd) <= 1e-5
    k2, kd2, kdd2 = (t.cpu().numpy() for t in sk.compute_kernel_and_derivatives_Gram(X.cuda(), Y.cuda(), (2 * g).cuda()))
    assert rel_err(k2, k) <= 1e-13
    assert rel_err(kd2, 2 * kd) <= 1e-8
    assert rel_err(kdd2, 4 * kdd) <= 1e-4


# ---------------------------------------------------------------------------------------------
# the fused derivative solver (sk_solve_deriv_static_f64, csrc/sk_wave_deriv_fused.hip)
# ---------------------------------------------------------------------------------------------
FUSED_DERIV_SHAPES = [(1, 3, 2, 129, 128, 5), (1, 2, 2, 65, 200, 8), (0, 2, 3, 193, 170, 7), (1, 3, 2, 130, 128, 5), (0, 2, 3, 100, 140, 8), (2, 2, 2, 70, 127, 3), (1, 2, 3, 128, 158, 16), (1, 4, 3, 50, 127, 4),
                      (0, 3, 2, 64, 126, 7), (1, 3, 4, 40, 170, 3), (0, 2, 3, 70, 200, 8), (1, 3, 2, 130, 180, 5), (2, 2, 2, 20, 161, 2),
                      (1, 2, 2, 65, 300, 12), (0, 5, 7, 129, 165, 4), (2, 2, 3, 70, 170, 9), (1, 2, 2, 193, 150, 3)]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["linear", "rbf"])
def test_gpu_fused_derivative_solver_is_bit_identical_to_the_unfused_route(kind, monkeypatch):
    """Static kernel, finite differences, increments and the three-state sweep in ONE kernel: the increments are formed in the
    operand order of sk_static_deriv_increments_* and swept with the stencil of sk_solve_deriv_*'s fast kernel, so the three outputs
    are BIT-identical to the unfused route -- one band and several, the boundary through L2 (rows of >= 80 units) and through the LDS
    ring (64 <= units < 80, not a multiple of 32 included), the shifted and the unshifted band layout, dyadic 0..2 -- and within its scope
    (path dim <= 8) sk_static_deriv_increments is never called (nothing of size pairs x M x N in HBM)."""
    import sigkernel_amd
    from sigkernel_amd import _lib
    be = _lib.get_backend()
    kern = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(0.8)
    for d, A, B, M, N, D in FUSED_DERIV_SHAPES:
        gen = torch.Generator().manual_seed(M + N)
        X, Y = walk(gen, A, M, D).cuda(), walk(gen, B, N, D).cuda()
        g = torch.randn(A, M, D, generator=gen, dtype=torch.float64).cuda()
        sk = sigkernel_amd.SigKernel(kern, d)
        monkeypatch.setattr(sigkernel_amd.routes, "no_fused_deriv", True)
        want = sk.compute_kernel_and_derivatives_Gram(X, Y, g)
        monkeypatch.setattr(sigkernel_amd.routes, "no_fused_deriv", False)
        with monkeypatch.context() as m:
            if D <= 8:      # (the fused solver's scope; wider paths keep the unfused route)
                m.setattr(type(be), "static_deriv_increments", lambda self, *a, **k: (_ for _ in ()).throw(AssertionError("increments materialised")))
            got = sk.compute_kernel_and_derivatives_Gram(X, Y, g)
        for a, b, name in zip(got, want, ("k", "k_gamma", "k_gamma_gamma")):
            assert torch.equal(a, b), ((d, A, B, M, N, D), name, float((a - b).abs().max()))


@pytest.mark.gpu
def test_gpu_fused_derivative_solver_against_the_oracle():
    """The same route against the CPU oracle's k_kgrad (the reference's stencil and pre-processing): two bands, LDS boundary.
    k_gamma at 1e-8 here: a last-bit difference between the device's exp and libm's in an RBF node is amplified by 1/eps = 1e4 (the
    module docstring's argument; 1e-9 holds for the linear kernel)."""
    import sigkernel_amd
    from oracle import oracle as O
    gen = torch.Generator().manual_seed(19)
    X, Y = walk(gen, 6, 100, 4), walk(gen, 5, 130, 4)
    g = torch.randn(6, 100, 4, generator=gen, dtype=torch.float64) / 8
    for kern in (sigkernel_amd.LinearKernel(), sigkernel_amd.RBFKernel(0.9)):
        k, kd, kdd = (t.cpu().numpy() for t in sigkernel_amd.SigKernel(kern, 1).compute_kernel_and_derivatives_Gram(X.cuda(), Y.cuda(), g.cuda()))
        ek, ekd, ekdd = O.kgrad(X, Y, g, kern, 1, nthreads=8)
        tol_kd = TOL_KD if isinstance(kern, sigkernel_amd.LinearKernel) else 1e-8
        assert rel_err(k, ek) <= TOL_K and rel_err(kd, ekd) <= tol_kd and rel_err(kdd, ekdd) <= 1e-5, (rel_err(k, ek), rel_err(kd, ekd), rel_err(kdd, ekdd))
