"""Statistics built on the signature Gram matrices (the reference's sigkernel.py:618-691).

Pure torch on top of SigKernel.compute_Gram / compute_mmd: nothing here touches the HIP kernels directly, it exists
so that users of the reference find the same entry points.
"""
import math

import torch

from .sigkernel import SigKernel

__all__ = ["c_alpha", "hypothesis_test", "SigCHSIC"]


def c_alpha(m, alpha):
    """Acceptance threshold of the MMD two-sample test for sample size m (sigkernel.py:618-619)."""
    return 4. * math.sqrt(-math.log(alpha) / m)


def hypothesis_test(y_pred, y_test, static_kernel, confidence_level=0.99, dyadic_order=0, verbose=True):
    """MMD two-sample test between two sets of paths (sigkernel.py:621-640).

    Prints the reference's verdict line; additionally returns (rejected, statistic, threshold)."""
    k_sig = SigKernel(static_kernel, dyadic_order)
    m = max(y_pred.shape[0], y_test.shape[0])
    stat = k_sig.compute_mmd(y_pred, y_test)
    thr = torch.tensor(c_alpha(m, confidence_level), dtype=y_pred.dtype, device=stat.device)
    rejected = bool(stat > thr)
    if verbose:
        if rejected:
            print(f'Hypothesis rejected: distribution are not equal with {confidence_level*100}% confidence')
        else:
            print(f'Hypothesis accepted: distribution are equal with {confidence_level*100}% confidence')
    return rejected, stat, thr


def SigCHSIC(X, Y, Z, static_kernel, dyadic_order=1, eps=0.1):
    """Signature conditional HSIC of X and Y given Z, each (batch, length, dim) (sigkernel.py:644-691).

    Follows the reference step by step, including its call of ``torch.cholesky_inverse`` on the regularised matrix
    itself (not on a Cholesky factor): parity with the reference means reproducing that."""
    m = X.shape[0]
    eye = torch.eye(m, dtype=X.dtype, device=X.device)
    H = eye - torch.full((m, m), 1. / m, dtype=X.dtype, device=X.device)          # centring matrix
    sk = SigKernel(static_kernel, dyadic_order)
    Kx, Ky, Kz = (H @ sk.compute_Gram(V, V, sym=True) @ H for V in (X, Y, Z))
    Kz_reg_inv = torch.cholesky_inverse(Kz + m * eps * eye)
    A = Kz @ (Kz_reg_inv @ Kz_reg_inv) @ Kz
    Bm = Kx @ A @ Ky
    return (torch.trace(Kx @ Ky) - 2. * torch.trace(Bm) + torch.trace(Bm @ A)) / m ** 2
