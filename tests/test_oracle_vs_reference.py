"""The oracle against the REAL reference, live (container only).

Skipped wherever /root/reference is absent (the GPU box): there the committed fixtures under tests/golden/ -- produced
by the same reference through tests/golden/make_golden.py -- stand in.  Here the reference's Cython solver is built
where it lies (oracle/build_ref.py, outputs under oracle/_ref/, never committed, never shipped) and driven with fresh
random inputs, so the pin does not rest on the fixtures alone.
"""
import os

import numpy as np
import pytest
import torch

REF = os.environ.get("SIGKERNEL_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "sigkernel")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    from oracle.build_ref import import_reference
    return import_reference()


def _walk(gen, A, M, D):
    return torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)


@pytest.mark.parametrize("naive", [False, True])
def test_solver_bit_identical_on_fresh_random_increments(ref, naive):
    from cython_backend import sigkernel_cython, sigkernel_Gram_cython
    from oracle import oracle as O
    rng = np.random.default_rng(20240917 + naive)
    for shape in [(4, 11, 6), (2, 5, 23), (1, 1, 1)]:
        inc = rng.normal(scale=0.4, size=shape)
        assert np.array_equal(O.solve_fine(inc, naive), sigkernel_cython(inc, naive))
    inc4 = rng.normal(scale=0.4, size=(3, 2, 7, 9))
    assert np.array_equal(O.solve_fine(inc4, naive), sigkernel_Gram_cython(inc4, False, naive))
    incs = rng.normal(scale=0.4, size=(3, 3, 6, 6))
    assert np.array_equal(O.gram_sym_fine(incs, naive), sigkernel_Gram_cython(incs, True, naive))


@pytest.mark.parametrize("kind,d", [("linear", 0), ("linear", 2), ("rbf", 1), ("rbf", 3)])
def test_gram_and_gradient_against_the_live_reference(ref, kind, d):
    import sigkernel_amd
    from oracle import oracle as O
    gen = torch.Generator().manual_seed(99 + d)
    X, Y = _walk(gen, 4, 9, 3) * 2, _walk(gen, 3, 12, 3) * 2
    rk = ref.LinearKernel() if kind == "linear" else ref.RBFKernel(0.8)
    ok = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(0.8)
    rs = ref.SigKernel(rk, dyadic_order=d)
    Xg = X.clone().requires_grad_(True)
    K = rs.compute_Gram(Xg, Y)
    w = torch.randn(4, 3, generator=gen, dtype=torch.float64)
    (K * w).sum().backward()
    got = O.gram_forward(X, Y, ok, d)
    assert np.max(np.abs(got - K.detach().numpy())) <= 1e-13 * np.max(np.abs(got))
    gp = O.gram_grad_points(X, Y, ok, d)
    grad = np.einsum("ab,abmd->amd", w.numpy(), gp)
    # the reference differentiates by forward differences with h = 1e-9 in double: measure ITS round-off noise against its own
    # formula in long double (tests/ld_reference.py); the analytic adjoint must sit within 1e-7 of that formula, and the
    # live reference within its measured noise of it
    from ld_reference import reference_gradient_ld, rel_err_ld
    ld = reference_gradient_ld(dict(X=X.numpy(), Y=Y.numpy(), w=w.numpy(), dyadic=d, naive=0), "grad_w",
                               kernel=kind, param=0.8)
    assert rel_err_ld(grad, ld) <= 1e-7
    noise = rel_err_ld(Xg.grad.numpy(), ld)
    assert noise <= 2e-5       # sanity bound on the reference's own noise, not a parity tolerance
    assert np.max(np.abs(grad - Xg.grad.numpy())) <= (noise + 1e-7) * np.max(np.abs(grad)) * 1.01


def test_derivative_stencil_against_the_live_reference(ref):
    from sigkernel.mps_backend import sigkernel_derivatives_Gram_mps
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    inc, inc_d, inc_dd = (torch.tensor(rng.normal(scale=0.4, size=(2, 2, 5, 8))) for _ in range(3))
    K = torch.zeros(2, 2, 6, 9, dtype=torch.float64)
    Kd, Kdd = torch.zeros_like(K), torch.zeros_like(K)
    K[:, :, 0, :] = 1.
    K[:, :, :, 0] = 1.
    sigkernel_derivatives_Gram_mps(inc, inc_d, inc_dd, 5, 8, K, Kd, Kdd)
    k, kd, kdd, grids = O.solve_deriv_coarse(inc.numpy(), inc_d.numpy(), inc_dd.numpy(), 0, want_grid=True)
    assert np.array_equal(grids[0], K.numpy()) and np.array_equal(grids[1], Kd.numpy()) and np.array_equal(grids[2], Kdd.numpy())
