"""Static kernels feeding the signature PDE (the reference's sigkernel/static_kernels.py:11-73).

Duck-typed like the reference: anything with ``batch_kernel(X, Y) -> (A, M, N)`` and
``Gram_matrix(X, Y) -> (A, B, M, N)`` works with :class:`sigkernel_amd.SigKernel`.
"""
import torch

__all__ = ["LinearKernel", "RBFKernel"]


class LinearKernel:
    """Linear kernel k(x, y) = <x, y>."""

    def __init__(self, scale=1.0):
        self.scale = scale

    def batch_kernel(self, X, Y):
        """(A,M,D), (A,N,D) -> (A,M,N); applies scale to both arguments (static_kernels.py:24)."""
        return torch.bmm(self.scale * X, (self.scale * Y).transpose(1, 2))

    def Gram_matrix(self, X, Y):
        """(A,M,D), (B,N,D) -> (A,B,M,N); like the reference this ignores ``scale`` (static_kernels.py:33).

        Same contraction as the reference's ``einsum('ipk,jqk->ijpq')``, issued as one broadcast batched GEMM so
        that the result is produced directly in (A,B,M,N) order (no 34 GB transpose copy at the headline size)."""
        return torch.matmul(X[:, None], Y[None].transpose(-1, -2))


class RBFKernel:
    """RBF kernel k(x, y) = exp(-|x - y|^2 / sigma)  (sigma, not 2 sigma^2: static_kernels.py:56,73)."""

    def __init__(self, sigma):
        self.sigma = sigma

    def batch_kernel(self, X, Y):
        A, M, N = X.shape[0], X.shape[1], Y.shape[1]
        Xs = torch.sum(X ** 2, dim=2)
        Ys = torch.sum(Y ** 2, dim=2)
        dist = -2. * torch.bmm(X, Y.permute(0, 2, 1))
        dist = dist + (torch.reshape(Xs, (A, M, 1)) + torch.reshape(Ys, (A, 1, N)))  # `dist += a + b` in the reference
        return torch.exp(-dist / self.sigma)

    def Gram_matrix(self, X, Y):
        A, B, M, N = X.shape[0], Y.shape[0], X.shape[1], Y.shape[1]
        Xs = torch.sum(X ** 2, dim=2)
        Ys = torch.sum(Y ** 2, dim=2)
        dist = -2. * torch.matmul(X[:, None], Y[None].transpose(-1, -2))   # einsum('ipk,jqk->ijpq') in (A,B,M,N) order
        dist = dist + (torch.reshape(Xs, (A, 1, M, 1)) + torch.reshape(Ys, (1, B, 1, N)))
        return torch.exp(-dist / self.sigma)
