"""ctypes front-end of oracle/libsk_oracle.so (see sigkernel_oracle.c).

TEST INFRASTRUCTURE ONLY -- never imported by the product path.
All functions take / return numpy float64 arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsk_oracle.so")
_lib = None

__all__ = ["build", "solve_fine", "gram_sym_fine", "solve_coarse", "adjoint_coarse",
           "increments", "increments_adjoint", "max_threads", "gram_forward", "gram_grad_points",
           "gram_grad_weighted", "solve_deriv_coarse", "kgrad"]


def build(force=False):
    src = os.path.join(_HERE, "sigkernel_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsk_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        d = ctypes.POINTER(ctypes.c_double)
        lib.sk_oracle_solve_fine.argtypes = [d, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, d]
        lib.sk_oracle_solve_fine.restype = None
        lib.sk_oracle_gram_sym_fine.argtypes = [d, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, d]
        lib.sk_oracle_gram_sym_fine.restype = None
        lib.sk_oracle_solve_coarse.argtypes = [d, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, d, d, ctypes.c_int]
        lib.sk_oracle_solve_coarse.restype = ctypes.c_int
        lib.sk_oracle_adjoint_coarse.argtypes = [d, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, d, d, ctypes.c_int]
        lib.sk_oracle_adjoint_coarse.restype = ctypes.c_int
        lib.sk_oracle_increments.argtypes = [d, ctypes.c_int64, ctypes.c_int, ctypes.c_int, d]
        lib.sk_oracle_increments.restype = None
        lib.sk_oracle_increments_adjoint.argtypes = [d, ctypes.c_int64, ctypes.c_int, ctypes.c_int, d]
        lib.sk_oracle_increments_adjoint.restype = None
        lib.sk_oracle_solve_deriv_coarse.argtypes = [d, d, d, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                     d, d, d, d, ctypes.c_int]
        lib.sk_oracle_solve_deriv_coarse.restype = ctypes.c_int
        lib.sk_oracle_max_threads.restype = ctypes.c_int
        lib.sk_oracle_gram_pipeline.argtypes = [d, ctypes.c_int64, ctypes.c_int, d, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, d, ctypes.c_int]
        lib.sk_oracle_gram_pipeline.restype = ctypes.c_int
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def max_threads():
    return int(_load().sk_oracle_max_threads())


def solve_fine(inc, naive=False):
    """inc [..., MM, NN] fine increments -> full grids [..., MM+1, NN+1] (cython_backend.pyx:7-33, :98-117)."""
    inc = _c(inc)
    MM, NN = inc.shape[-2:]
    P = int(np.prod(inc.shape[:-2], dtype=np.int64))
    K = np.empty(inc.shape[:-2] + (MM + 1, NN + 1), dtype=np.float64)
    _load().sk_oracle_solve_fine(_p(inc), P, MM, NN, int(naive), _p(K))
    return K


def gram_sym_fine(inc, naive=False):
    """inc [A, A, MM, NN] -> [A, A, MM+1, NN+1], sym=True branch (cython_backend.pyx:74-97)."""
    inc = _c(inc)
    A, A2, MM, NN = inc.shape
    assert A == A2
    K = np.empty((A, A, MM + 1, NN + 1), dtype=np.float64)
    _load().sk_oracle_gram_sym_fine(_p(inc), A, MM, NN, int(naive), _p(K))
    return K


def solve_coarse(inc_c, dyadic, naive=False, want_grid=False, nthreads=1):
    """inc_c [..., Mc, Nc] coarse increments -> final values [...] (and optionally the full grids)."""
    inc_c = _c(inc_c)
    Mc, Nc = inc_c.shape[-2:]
    P = int(np.prod(inc_c.shape[:-2], dtype=np.int64))
    out = np.empty(inc_c.shape[:-2], dtype=np.float64)
    grid = None
    if want_grid:
        grid = np.empty(inc_c.shape[:-2] + ((Mc << dyadic) + 1, (Nc << dyadic) + 1), dtype=np.float64)
    rc = _load().sk_oracle_solve_coarse(_p(inc_c), P, Mc, Nc, int(dyadic), int(naive), _p(out),
                                        _p(grid) if grid is not None else None, int(nthreads))
    if rc:
        raise RuntimeError("sk_oracle_solve_coarse failed (%d)" % rc)
    return (out, grid) if want_grid else out


def adjoint_coarse(inc_c, dyadic, naive=False, nthreads=1):
    """inc_c [..., Mc, Nc] -> (final [...], W [..., Mc, Nc] = d k_sig / d inc_c)."""
    inc_c = _c(inc_c)
    Mc, Nc = inc_c.shape[-2:]
    P = int(np.prod(inc_c.shape[:-2], dtype=np.int64))
    out = np.empty(inc_c.shape[:-2], dtype=np.float64)
    W = np.empty_like(inc_c)
    rc = _load().sk_oracle_adjoint_coarse(_p(inc_c), P, Mc, Nc, int(dyadic), int(naive), _p(out), _p(W), int(nthreads))
    if rc:
        raise RuntimeError("sk_oracle_adjoint_coarse failed (%d)" % rc)
    return out, W


def increments(G):
    """G [..., M, N] static Gram -> inc_c [..., M-1, N-1] (sigkernel.py:217, :363)."""
    G = _c(G)
    M, N = G.shape[-2:]
    P = int(np.prod(G.shape[:-2], dtype=np.int64))
    out = np.empty(G.shape[:-2] + (M - 1, N - 1), dtype=np.float64)
    _load().sk_oracle_increments(_p(G), P, M, N, _p(out))
    return out


def increments_adjoint(W):
    """W [..., M-1, N-1] = dL/dinc_c -> dL/dG [..., M, N]."""
    W = _c(W)
    Mc, Nc = W.shape[-2:]
    P = int(np.prod(W.shape[:-2], dtype=np.int64))
    out = np.empty(W.shape[:-2] + (Mc + 1, Nc + 1), dtype=np.float64)
    _load().sk_oracle_increments_adjoint(_p(W), P, Mc + 1, Nc + 1, _p(out))
    return out


def solve_deriv_coarse(inc_c, incd_c, incdd_c, dyadic, want_grid=False, nthreads=1):
    """Three coarse increment arrays [..., Mc, Nc] -> (k, k_gamma, k_gamma_gamma) [...] each
    (cuda_backend.py:166-223); with want_grid also the three full grids [3, ..., MM+1, NN+1]."""
    inc_c, incd_c, incdd_c = _c(inc_c), _c(incd_c), _c(incdd_c)
    assert inc_c.shape == incd_c.shape == incdd_c.shape
    Mc, Nc = inc_c.shape[-2:]
    P = int(np.prod(inc_c.shape[:-2], dtype=np.int64))
    outs = [np.empty(inc_c.shape[:-2], dtype=np.float64) for _ in range(3)]
    grids = None
    if want_grid:
        grids = np.empty((3,) + inc_c.shape[:-2] + ((Mc << dyadic) + 1, (Nc << dyadic) + 1), dtype=np.float64)
    rc = _load().sk_oracle_solve_deriv_coarse(_p(inc_c), _p(incd_c), _p(incdd_c), P, Mc, Nc, int(dyadic), _p(outs[0]),
                                              _p(outs[1]), _p(outs[2]), _p(grids) if grids is not None else None,
                                              int(nthreads))
    if rc:
        raise RuntimeError("sk_oracle_solve_deriv_coarse failed (%d)" % rc)
    return tuple(outs) + ((grids,) if want_grid else ())


# ---------------------------------------------------------------------------
# End-to-end restatements of the reference's autograd functions, composed from
# the C pieces above plus torch for the static kernel (the reference also uses
# torch there, so parity at that boundary is by construction).
# ---------------------------------------------------------------------------
def gram_forward(X, Y, static_kernel, dyadic, naive=False, nthreads=1):
    """_SigKernelGram.forward (sigkernel.py:350-401): torch (A,M,D),(B,N,D) -> numpy (A,B)."""
    G = static_kernel.Gram_matrix(X.detach().double().cpu(), Y.detach().double().cpu()).numpy()
    return solve_coarse(increments(G), dyadic, naive, nthreads=nthreads)


def gram_pipeline(X, Y, kind, param, dyadic, naive=False, nthreads=1):
    """The all-cores CPU baseline of bench.py: static kernel (kind 0 linear / 1 rbf with sigma = param), increments and PDE
    solve per pair inside one OpenMP region (sk_oracle_gram_pipeline).  numpy (A,M,D), (B,N,D) -> (A,B)."""
    X, Y = _c(X), _c(Y)
    A, M, D = X.shape
    B, N = Y.shape[0], Y.shape[1]
    out = np.empty((A, B), dtype=np.float64)
    rc = _load().sk_oracle_gram_pipeline(_p(X), A, M, _p(Y), B, N, D, int(kind), float(param), int(dyadic), int(naive), _p(out),
                                         int(nthreads))
    if rc:
        raise RuntimeError("sk_oracle_gram_pipeline failed (%d)" % rc)
    return out


def gram_grad_points(X, Y, static_kernel, dyadic, naive=False, nthreads=1):
    """prep_backward (sigkernel.py:419-502) in closed form with an analytic static-kernel derivative.

    Returns numpy (A,B,M,D): d k_sig(x_a, y_b) / d x_a[m,:] (Y held constant).
    """
    import torch
    Xd = X.detach().double().cpu().requires_grad_(True)
    Yd = Y.detach().double().cpu()
    A, M, D = Xd.shape
    B = Yd.shape[0]
    with torch.enable_grad():
        G = static_kernel.Gram_matrix(Xd, Yd)
    _, W = adjoint_coarse(increments(G.detach().numpy()), dyadic, naive, nthreads=nthreads)
    dG = torch.from_numpy(increments_adjoint(W))
    out = np.empty((A, B, M, D))
    for b in range(B):
        mask = torch.zeros_like(dG)
        mask[:, b] = dG[:, b]
        (g,) = torch.autograd.grad(G, Xd, grad_outputs=mask, retain_graph=True)
        out[:, b] = g.numpy()
    return out


def gram_grad_weighted(X, Y, w, static_kernel, dyadic, naive=False, nthreads=1):
    """_SigKernelGram.backward for an upstream gradient w (A,B) (sigkernel.py:404-416 without the 2x rule):
    numpy (A,M,D) = sum_b w[a,b] * gram_grad_points[a,b] -- one vector-Jacobian product instead of B of them, for the
    batch sizes where the per-pair tensor (A,B,M,D) would not fit."""
    import torch
    Xd = X.detach().double().cpu().requires_grad_(True)
    Yd = Y.detach().double().cpu()
    with torch.enable_grad():
        G = static_kernel.Gram_matrix(Xd, Yd)
    _, W = adjoint_coarse(increments(G.detach().numpy()), dyadic, naive, nthreads=nthreads)
    dG = torch.from_numpy(increments_adjoint(W)) * torch.as_tensor(np.asarray(w, dtype=np.float64))[:, :, None, None]
    (g,) = torch.autograd.grad(G, Xd, grad_outputs=dG)
    return g.numpy()


def kgrad(X, Y, gamma, static_kernel, dyadic, eps=1e-4, nthreads=1):
    """k_kgrad (sigkernel.py:504-593): (k, k_gamma, k_gamma_gamma), numpy (A,B) each.

    The three increment arrays are the 4-corner differences of the scaled static Gram matrices of
    X, X + eps*gamma and X + 2*eps*gamma, combined in the reference's order (sigkernel.py:526-541)."""
    Xd, Yd, gd = (t.detach().double().cpu() for t in (X, Y, gamma))
    G0 = static_kernel.Gram_matrix(Xd, Yd)
    d1 = -(1. / eps) * G0
    d2 = (1. / eps) * static_kernel.Gram_matrix(Xd + eps * gd, Yd)
    dd1 = -(1. / eps) * d1
    dd2 = -(2. / eps) * d2
    dd3 = (1. / eps ** 2) * static_kernel.Gram_matrix(Xd + 2. * eps * gd, Yd)
    inc = increments(G0.numpy())
    inc_d = increments(d1.numpy()) + increments(d2.numpy())
    inc_dd = increments(dd1.numpy()) + increments(dd2.numpy()) + increments(dd3.numpy())
    return solve_deriv_coarse(inc, inc_d, inc_dd, dyadic, nthreads=nthreads)
