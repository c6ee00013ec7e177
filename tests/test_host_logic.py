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
