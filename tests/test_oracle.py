"""Pins the CPU oracle (oracle/sigkernel_oracle.c) to the golden vectors produced by the real
reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import scipy.special
import torch

from conftest import GOLDEN, golden, golden_gram_cases, make_kernel, rel_err, grad_tol
from oracle import oracle as O

FWD_TOL = 1e-13   # the oracle is the same arithmetic as the reference: expect exactly 0
# The reference differentiates the static kernel by forward differences with h = 1e-9 in fp64
# (sigkernel.py:472-487): its own gradients carry rounding noise of 1e-7..7e-6 (max-norm relative;
# largest on the rough RBF cases).  test_adjoint_vs_noise_free_reference_formula shows the noise is the
# reference's, not ours: against the same formula evaluated in extended precision we agree to 1e-8.


def test_solver_grids_bit_identical():
    g = golden("solver_grids")
    for naive in (0, 1):
        assert np.array_equal(O.solve_fine(g["inc3"], naive), g["grid3_naive%d" % naive])
        assert np.array_equal(O.solve_fine(g["inc4"], naive), g["grid4_naive%d" % naive])
        assert np.array_equal(O.gram_sym_fine(g["incs"], naive), g["grids_sym_naive%d" % naive])


def test_coarse_solver_equals_fine_solver():
    rng = np.random.default_rng(0)
    inc_c = rng.normal(scale=0.2, size=(4, 5, 7))
    for d in (0, 1, 2, 3):
        r = 1 << d
        fine = np.repeat(np.repeat(inc_c, r, axis=1) / r, r, axis=2) / r   # tile(tile(.)/r)/r, sigkernel.py:218
        for naive in (0, 1):
            grid = O.solve_fine(fine, naive)
            out, grid_c = O.solve_coarse(inc_c, d, naive, want_grid=True)
            assert np.array_equal(grid, grid_c)
            assert np.array_equal(out, grid[:, -1, -1])
            assert np.array_equal(O.solve_coarse(inc_c, d, naive, nthreads=4), out)


@pytest.mark.parametrize("name", golden_gram_cases())
def test_gram_forward_matches_reference(name):
    c = golden(name)
    X, Y = torch.from_numpy(c["X"]), torch.from_numpy(c["Y"])
    K = O.gram_forward(X, Y, make_kernel(c), int(c["dyadic"]), bool(c["naive"]))
    assert rel_err(K, c["gram"]) <= FWD_TOL
    assert rel_err(c["gram_tiled"], c["gram"]) <= FWD_TOL   # reference property: max_batch does not change results
    n = c["paired"].shape[0]
    assert rel_err(np.diag(K[:n, :n]), c["paired"]) <= 1e-12


@pytest.mark.parametrize("name", golden_gram_cases())
def test_adjoint_matches_reference_gradients(name):
    c = golden(name)
    X, Y = torch.from_numpy(c["X"]), torch.from_numpy(c["Y"])
    gp = O.gram_grad_points(X, Y, make_kernel(c), int(c["dyadic"]), bool(c["naive"]))   # (A,B,M,D)
    grad = np.einsum("ab,abmd->amd", c["w"], gp)
    assert rel_err(grad, c["grad_w"]) <= grad_tol(name, "grad_w")
    if "grad_xx_sum" in c:
        gp_xx = O.gram_grad_points(X, X, make_kernel(c), int(c["dyadic"]), bool(c["naive"]))
        assert rel_err(2 * gp_xx.sum(axis=1), c["grad_xx_sum"]) <= grad_tol(name, "grad_xx_sum")   # the 2x rule, sigkernel.py:410-412


def test_adjoint_weights_are_the_exact_derivative_of_the_surrogate():
    # W = d/d inc_c of sum(KK * inc) with KK frozen: check against its definition on a tiny grid
    rng = np.random.default_rng(3)
    inc_c = rng.normal(scale=0.3, size=(2, 3, 4))
    for d in (0, 1, 2):
        r = 1 << d
        out, W = O.adjoint_coarse(inc_c, d)
        fine = np.repeat(np.repeat(inc_c, r, axis=1) / r, r, axis=2) / r
        K = O.solve_fine(fine)
        Kr = O.solve_fine(fine[:, ::-1, ::-1].copy())[:, ::-1, ::-1]
        KK = K[:, :-1, :-1] * Kr[:, 1:, 1:]                                   # sigkernel.py:469-470
        Wref = KK.reshape(2, 3, r, 4, r).sum(axis=(2, 4)) / r / r
        assert rel_err(W, Wref) <= 1e-14
        assert np.array_equal(out, K[:, -1, -1])


def test_increments_and_adjoint_are_transposes():
    rng = np.random.default_rng(5)
    G = rng.normal(size=(3, 6, 5))
    W = rng.normal(size=(3, 5, 4))
    lhs = np.sum(O.increments(G) * W)
    rhs = np.sum(G * O.increments_adjoint(W))
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs)


def test_straight_line_known_answers():
    """<dx,dy> = c = 1: d=0 gives exactly 2.25 and the scheme converges to I0(2)."""
    k = golden("kat_straight_lines")
    inc_c = np.ones((1, 1, 1))
    assert O.solve_coarse(inc_c, 0)[0] == 2.25 == k["d0"][0]
    for d in (1, 4, 8):
        assert O.solve_coarse(inc_c, d)[0] == k["d%d" % d][0]
    assert abs(O.solve_coarse(inc_c, 8)[0] - scipy.special.i0(2.0)) < 1e-6


def test_readme_example_values():
    c = golden("readme_c1")
    import sigkernel_amd
    X, Y = torch.from_numpy(c["X"]), torch.from_numpy(c["Y"])
    K = O.gram_forward(X, Y, sigkernel_amd.RBFKernel(float(c["sigma"])), int(c["dyadic"]))
    assert rel_err(K, c["gram"]) <= FWD_TOL
    assert rel_err(np.diag(K), c["kernel"]) <= 1e-12


def _fixture_gradient_keys():
    import json
    with open(os.path.join(GOLDEN, "grad_errors.json")) as f:
        table = json.load(f)
    return [(n, k) for n in sorted(table) if not n.startswith("_") for k in sorted(table[n])]


def _oracle_gradient(c, key, kernel):
    """The analytic adjoint (oracle closed form) combined like the reference combines grad_points for fixture gradient `key`."""
    d, naive = int(c["dyadic"]), bool(c["naive"]) if "naive" in c else False
    X, Y = torch.from_numpy(c["X"]), torch.from_numpy(c["Y"])
    A, B = X.shape[0], Y.shape[0]
    if key == "grad_w":
        return O.gram_grad_weighted(X, Y, c["w"], kernel, d, naive)
    if key == "grad_xx_sum":
        return 2 * O.gram_grad_weighted(X, X, np.ones((A, A)), kernel, d, naive)
    if key == "grad_mmd":
        wxx = (np.ones((A, A)) - np.eye(A)) / (A * (A - 1.0))
        return 2 * O.gram_grad_weighted(X, X, wxx, kernel, d, naive) + O.gram_grad_weighted(X, Y, np.full((A, B), -2.0 / (A * B)), kernel, d, naive)
    n = c["wp"].shape[0] if key == "grad_paired" else A
    wp = c["wp"] if key == "grad_paired" else np.ones(A)
    gp = O.gram_grad_points(X[:n], Y[:n], kernel, d, naive)       # paired = the diagonal of the Gram of the first n paths
    return wp[:, None, None] * gp[np.arange(n), np.arange(n)]


def test_the_gradient_yardstick_is_independent_of_what_it_judges():
    """tests/ld_reference.py (static kernel, PDE solutions K and K~, the weights 4^-d sum_cell K K~ -- all in long double by plain
    numpy loops) and tests/golden/measure_grad_noise.py import neither the oracle, nor the product, nor the reference; and its PDE
    weights agree with the oracle's to double precision, which pins the ORACLE's adjoint against an independent evaluation."""
    import ld_reference
    for path in (ld_reference.__file__, os.path.join(os.path.dirname(ld_reference.__file__), "golden", "measure_grad_noise.py")):
        src = open(path).read()
        assert "import oracle" not in src and "from oracle" not in src and "import sigkernel" not in src and "from sigkernel" not in src
    rng = np.random.default_rng(5)
    for d, naive in ((0, False), (1, False), (2, False), (1, True)):
        inc = rng.normal(size=(3, 5, 7)) * 0.3
        W_ld = ld_reference.adjoint_weights_ld(inc, d, naive)
        _, W = O.adjoint_coarse(inc, d, naive)
        assert float(np.max(np.abs(W - W_ld.astype(np.float64))) / np.max(np.abs(W))) <= 1e-13


@pytest.mark.parametrize("name,key", _fixture_gradient_keys())
def test_every_gradient_fixture_vs_noise_free_reference_formula(name, key):
    """prep_backward's formula (sigkernel.py:469-500; paired :313-341) with its h = 1e-9 forward difference evaluated in long
    double (tests/ld_reference.py), so that only the O(h) truncation error is left.  Per (fixture, gradient):
    (i) the analytic adjoint matches that formula to 1e-7; (ii) the fixture's distance from it -- the reference's own
    round-off noise -- is what tests/golden/grad_errors.json records, and conftest.grad_tol is derived from THAT number (never
    from what an implementation achieves)."""
    import sigkernel_amd
    from ld_reference import reference_gradient_ld, rel_err_ld
    from conftest import reference_noise
    c = golden(name)
    if name == "readme_c1":
        c = dict(c, kernel="rbf", param=float(c["sigma"]), naive=0)
    kernel = sigkernel_amd.LinearKernel() if str(c["kernel"]) == "linear" else sigkernel_amd.RBFKernel(float(c["param"]))
    ld = reference_gradient_ld(c, key)
    assert rel_err_ld(_oracle_gradient(c, key, kernel), ld) <= 1e-7                     # (i)
    noise = rel_err_ld(c[key], ld)                                                      # (ii)
    assert abs(noise - reference_noise(name, key)) <= 1e-3 * reference_noise(name, key) + 1e-12
    assert noise + 1e-7 <= grad_tol(name, key)
