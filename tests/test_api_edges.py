"""The reference API's corner inputs on the GPU: strided views, path dimensions beyond every fused kernel's limit,
one-pair batches of two-point paths, duck-typed static kernels, dyadic orders past the wavefront kernels' range.
Every case goes through SigKernel (sigkernel/sigkernel.py:77-229 in the reference) and is judged by the oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_err, walk

pytestmark = pytest.mark.gpu


def _api():
    import sigkernel_amd
    from oracle import oracle as O
    return sigkernel_amd, O


def _check_gram(kernel, d, X, Y, tol_k=1e-12, tol_g=1e-9):
    S, O = _api()
    sk = S.SigKernel(kernel, d)
    Xg = X.clone().requires_grad_(True)
    K = sk.compute_Gram(Xg, Y)
    K.sum().backward()
    Xc, Yc = X.detach().cpu().double().contiguous(), Y.detach().cpu().double().contiguous()
    assert rel_err(K.detach().cpu().numpy(), O.gram_forward(Xc, Yc, kernel, d)) <= tol_k
    assert rel_err(Xg.grad.cpu().numpy(), O.gram_grad_points(Xc, Yc, kernel, d).sum(1)) <= tol_g


def test_strided_views_are_accepted():
    S, _ = _api()
    gen = torch.Generator().manual_seed(0)
    X = walk(gen, 6, 20, 6).cuda()[:, ::2, ::2]
    Y = walk(gen, 5, 17, 3).cuda()
    assert not X.is_contiguous()
    _check_gram(S.RBFKernel(0.7), 1, X, Y)


@pytest.mark.parametrize("kind", ["rbf", "linear"])
def test_path_dimension_beyond_the_fused_kernels(kind):
    S, _ = _api()
    gen = torch.Generator().manual_seed(1)
    X, Y = walk(gen, 4, 9, 40).cuda(), walk(gen, 3, 11, 40).cuda()
    _check_gram(S.RBFKernel(2.0) if kind == "rbf" else S.LinearKernel(), 1, X, Y)


def test_single_pair_of_two_point_paths():
    S, O = _api()
    gen = torch.Generator().manual_seed(2)
    X, Y = walk(gen, 1, 2, 2).cuda(), walk(gen, 1, 2, 2).cuda()
    sk = S.SigKernel(S.LinearKernel(), 2)
    Xg = X.clone().requires_grad_(True)
    k = sk.compute_kernel(Xg, Y)
    k.sum().backward()
    assert abs(float(k.detach()) - O.gram_forward(X.cpu(), Y.cpu(), S.LinearKernel(), 2)[0, 0]) <= 1e-13
    assert torch.isfinite(Xg.grad).all()


def test_second_argument_must_not_require_grad():
    S, _ = _api()
    gen = torch.Generator().manual_seed(3)
    X, Y = walk(gen, 3, 5, 2).cuda(), walk(gen, 3, 5, 2).cuda().requires_grad_(True)
    with pytest.raises(AssertionError):            # sigkernel.py:177 in the reference
        S.SigKernel(S.LinearKernel(), 0).compute_mmd(X, Y)


def test_sym_flag_with_an_equal_copy():
    S, _ = _api()
    gen = torch.Generator().manual_seed(4)
    X = walk(gen, 10, 12, 3).cuda()
    sk = S.SigKernel(S.RBFKernel(1.0), 1)
    assert torch.allclose(sk.compute_Gram(X, X.clone(), sym=True), sk.compute_Gram(X, X.clone(), sym=False), atol=1e-13)


def test_fp32_very_unbalanced_lengths():
    S, _ = _api()
    gen = torch.Generator().manual_seed(5)
    X, Y = walk(gen, 3, 300, 2, torch.float32).cuda(), walk(gen, 4, 5, 2, torch.float32).cuda()
    _check_gram(S.RBFKernel(1.0), 0, X, Y, tol_k=1e-5, tol_g=1e-3)


class _Poly:
    """A user-supplied static kernel: only the two methods the reference calls (static_kernels.py:24-36)."""

    def batch_kernel(self, X, Y):
        return (1 + torch.bmm(X, Y.permute(0, 2, 1))) ** 2

    def Gram_matrix(self, X, Y):
        return (1 + torch.einsum('ipk,jqk->ijpq', X, Y)) ** 2


def test_duck_typed_static_kernel():
    gen = torch.Generator().manual_seed(6)
    X, Y = walk(gen, 4, 8, 3).cuda(), walk(gen, 3, 9, 3).cuda()
    _check_gram(_Poly(), 1, X, Y)


@pytest.mark.parametrize("d", [3, 4])
def test_dyadic_orders_past_the_wavefront_kernels(d):
    S, _ = _api()
    gen = torch.Generator().manual_seed(7)
    X, Y = walk(gen, 3, 6, 2).cuda(), walk(gen, 2, 7, 2).cuda()
    _check_gram(S.RBFKernel(1.0), d, X, Y)


@pytest.mark.parametrize("kind", ["linear", "rbf"])
def test_empty_batches_give_empty_results(kind):
    """The reference's CPU solver returns empty arrays for an empty batch (cython_backend.pyx:72 allocates (A,B,..) and the
    loops do not run); so does this, with a zero gradient of the right shape."""
    S, _ = _api()
    sk = S.SigKernel(S.LinearKernel() if kind == "linear" else S.RBFKernel(1.0), 1)
    X0 = torch.zeros(0, 8, 3, dtype=torch.float64, device="cuda")
    Y = torch.randn(4, 9, 3, dtype=torch.float64, device="cuda")
    assert sk.compute_Gram(X0, Y).shape == (0, 4) and sk.compute_Gram(Y, X0).shape == (4, 0)
    assert sk.compute_kernel(X0, X0).shape == (0,)
    Xg = X0.clone().requires_grad_(True)
    sk.compute_Gram(Xg, Y).sum().backward()
    assert Xg.grad.shape == X0.shape
    Yg = Y.clone().requires_grad_(True)
    sk.compute_Gram(Yg, X0).sum().backward()
    assert torch.equal(Yg.grad, torch.zeros_like(Y))
