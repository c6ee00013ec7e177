"""The reference's gradient FORMULA evaluated without its round-off noise (test infrastructure; imports neither the product
package, nor the oracle, nor the reference: numpy only -- the yardstick is independent of everything it judges).

The reference differentiates the static kernel by a forward difference with h = 1e-9 in double precision
(sigkernel.py:313-341 paired, :472-500 Gram): G_h - G_static cancels ~9 of 16 digits, so every gradient it returns -- and
every gradient fixture under tests/golden/ -- carries round-off noise of 1e-7 .. 1e-5 relative.  Here the same formula
(same h, same Diff_1 / Diff_2 / grad_points bookkeeping) is evaluated with the static kernel in numpy long double, so that
only the O(h) truncation error (~1e-9) is left.  That value is the yardstick for two separate statements:

  (i)  the analytic adjoint (oracle closed form, HIP kernels) agrees with the reference's formula to <= 1e-7;
  (ii) the distance of a fixture from it IS the reference's own noise -- the only thing a tolerance above north_star's 1e-6
       may rest on (tests/golden/grad_errors.json, written by tests/golden/measure_grad_noise.py).
"""
import numpy as np

LD = np.longdouble
H = LD(1e-9)      # sigkernel.py:313, :472


def _static_ld(kernel, param, Xa, Yb, gram):
    """Static kernel matrix in long double: gram -> (A,B,M,N) (static_kernels.py:26-33, :58-73), paired -> (A,M,N)
    (static_kernels.py:18-24, :40-56)."""
    Xa, Yb = Xa.astype(LD), Yb.astype(LD)
    if gram:
        xy = np.einsum("ipk,jqk->ijpq", Xa, Yb)
        xs, ys = (Xa ** 2).sum(2)[:, None, :, None], (Yb ** 2).sum(2)[None, :, None, :]
    else:
        xy = np.einsum("ipk,iqk->ipq", Xa, Yb)
        xs, ys = (Xa ** 2).sum(2)[:, :, None], (Yb ** 2).sum(2)[:, None, :]
    if kernel == "linear":
        return xy
    return np.exp(-(-2 * xy + xs + ys) / LD(param))


def _corner(G):
    return G[..., 1:, 1:] + G[..., :-1, :-1] - G[..., 1:, :-1] - G[..., :-1, 1:]


def _solve_fine_ld(inc, naive):
    """The reference's solver (cython_backend.pyx:101-117) on FINE increments [..., MM, NN] in long double, all pairs at once:
    the full grids [..., MM+1, NN+1]."""
    MM, NN = inc.shape[-2:]
    K = np.ones(inc.shape[:-2] + (MM + 1, NN + 1), dtype=LD)
    one, half, twelfth = LD(1), LD(0.5), LD(1) / LD(12)
    for i in range(MM):
        for j in range(NN):
            g = inc[..., i, j]
            if naive:                                       # cython_backend.pyx:114
                K[..., i + 1, j + 1] = (K[..., i + 1, j] + K[..., i, j + 1]) * (one + half * g) - K[..., i, j]
            else:                                           # cython_backend.pyx:116
                K[..., i + 1, j + 1] = (K[..., i + 1, j] + K[..., i, j + 1]) * (one + half * g + twelfth * g * g) - \
                    K[..., i, j] * (one - twelfth * g * g)
    return K


def adjoint_weights_ld(inc_c, dyadic, naive):
    """W[..., p, q] = 4^-d sum over the fine cells (i, j) of coarse cell (p, q) of K[i][j] K~[i+1][j+1] (sigkernel.py:438-470 +
    :489-495: the solution on the refined increments, the solution on the doubly flipped ones flipped back, their product
    folded over the r x r fine cells) in long double, by a plain double loop -- nothing borrowed from the oracle."""
    r = 1 << dyadic
    inc = np.repeat(np.repeat(inc_c.astype(LD), r, axis=-2), r, axis=-1) / LD(r * r)        # tile(): sigkernel.py:218, :364
    K = _solve_fine_ld(inc, naive)
    Kr = _solve_fine_ld(inc[..., ::-1, ::-1], naive)[..., ::-1, ::-1]                        # :438-469
    KK = K[..., :-1, :-1] * Kr[..., 1:, 1:]                                                   # :470
    Mc, Nc = inc_c.shape[-2:]
    return KK.reshape(KK.shape[:-2] + (Mc, r, Nc, r)).sum(axis=(-3, -1)) / LD(r * r)


def grad_points_ld(kernel, param, X, Y, dyadic, naive, gram=True):
    """grad_points of prep_backward (sigkernel.py:419-502; gram=False: _SigKernel.backward, :255-343) with the static
    kernel AND the PDE weights W = 4^-d sum_cell K K~ in long double: (A,B,M,D) for a Gram, (A,M,D) paired."""
    M, D = X.shape[1], X.shape[2]
    G0 = _static_ld(kernel, param, X, Y, gram)
    W = adjoint_weights_ld(_corner(G0), dyadic, bool(naive))
    out = np.zeros(G0.shape[:-2] + (M, D), dtype=LD)
    for k in range(D):
        Xh = X.astype(LD).copy()
        Xh[:, :, k] += H                                   # node (m, n) of G_h only sees x_m: all m perturbed at once
        dG = (_static_ld(kernel, param, Xh, Y, gram) - G0) / H
        # Diff_1 / h: d inc[p,q] / d x[p+1];  (Diff_2 - Diff_1) / h: d inc[p,q] / d x[p]
        T1 = (W * (dG[..., 1:, 1:] - dG[..., 1:, :-1])).sum(-1)
        T0 = (W * (dG[..., :-1, :-1] - dG[..., :-1, 1:])).sum(-1)
        out[..., :-1, k] += T0                              # grad_2 - grad_1 at row p
        out[..., 1:, k] += T1                               # grad_1 at row p + 1
    return out


def reference_gradient_ld(c, key, kernel=None, param=None):
    """The gradient `key` of golden fixture `c` (a dict of arrays) as the reference's formula gives it in long double,
    combined exactly like the reference combines grad_points for that fixture (tests/golden/make_golden.py)."""
    kernel = str(c["kernel"]) if kernel is None else kernel
    param = float(c["param"]) if param is None else param
    d, naive = int(c["dyadic"]), bool(c["naive"]) if "naive" in c else False
    X, Y = c["X"], c["Y"]
    A, B = X.shape[0], Y.shape[0]
    if key == "grad_w":                                    # (compute_Gram(X, Y) * w).sum()
        gp = grad_points_ld(kernel, param, X, Y, d, naive)
        g = np.einsum("ab,abmd->amd", c["w"].astype(LD), gp)
    elif key == "grad_paired":                             # (compute_kernel(X[:n], Y[:n]) * wp).sum()
        n = c["wp"].shape[0]
        g = c["wp"].astype(LD)[:, None, None] * grad_points_ld(kernel, param, X[:n], Y[:n], d, naive, gram=False)
    elif key == "grad_kernel_sum":                         # compute_kernel(X, Y).sum()
        g = grad_points_ld(kernel, param, X, Y, d, naive, gram=False)
    elif key == "grad_xx_sum":                             # compute_Gram(X, X, sym=True).sum(): the 2x rule, sigkernel.py:410-412
        g = 2 * grad_points_ld(kernel, param, X, X, d, naive).sum(axis=1)
    elif key == "grad_mmd":                                # compute_mmd (sigkernel.py:180-197): K_XX under the 2x rule, K_XY plain
        gxx = grad_points_ld(kernel, param, X, X, d, naive)
        gxy = grad_points_ld(kernel, param, X, Y, d, naive)
        wxx = ((np.ones((A, A)) - np.eye(A)) / (A * (A - 1.0))).astype(LD)
        g = 2 * np.einsum("ab,abmd->amd", wxx, gxx) - (LD(2) / (A * B)) * gxy.sum(axis=1)
    else:
        raise KeyError(key)
    return g


def rel_err_ld(a, b):
    a, b = np.asarray(a, dtype=LD), np.asarray(b, dtype=LD)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), LD(1e-300)))
