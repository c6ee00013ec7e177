"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Container-only: needs /root/reference (read-only) and builds its Cython solver
into oracle/_ref/ via oracle/build_ref.py. The reference never travels; only the
.npz files written here (inputs + expected outputs, pure data) are committed.

    python tests/golden/make_golden.py

Cases follow SURVEY.md section 8(c). Every array is float64.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.build_ref import import_reference  # noqa: E402

ref = import_reference()
from cython_backend import sigkernel_cython, sigkernel_Gram_cython  # noqa: E402  (built into oracle/_ref)


def walk(gen, A, M, D):
    """Scaled random walk, the bench input of SURVEY 8(d)."""
    return torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)


def kernel_of(name, param):
    return ref.LinearKernel() if name == "linear" else ref.RBFKernel(param)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------
# 1. raw solver: fine increments in, full grid out (cython_backend.pyx)
# ---------------------------------------------------------------------------
def solver_cases():
    rng = np.random.default_rng(1234)
    out = {}
    inc3 = rng.normal(scale=0.3, size=(3, 7, 9))
    inc4 = rng.normal(scale=0.3, size=(2, 3, 6, 6))
    incs = rng.normal(scale=0.3, size=(3, 3, 5, 5))
    out["inc3"] = inc3
    out["inc4"] = inc4
    out["incs"] = incs
    for naive in (0, 1):
        out["grid3_naive%d" % naive] = sigkernel_cython(inc3, bool(naive))
        out["grid4_naive%d" % naive] = sigkernel_Gram_cython(inc4, False, bool(naive))
        out["grids_sym_naive%d" % naive] = sigkernel_Gram_cython(incs, True, bool(naive))
    save("solver_grids", **out)


# ---------------------------------------------------------------------------
# 2. README example == BASELINE config 1 (README.md:37-81)
# ---------------------------------------------------------------------------
def readme_case():
    torch.manual_seed(0)
    X = torch.rand((5, 10, 2), dtype=torch.float64)
    Y = torch.rand((5, 20, 2), dtype=torch.float64)
    Z = torch.rand((3, 12, 2), dtype=torch.float64)
    sk = ref.SigKernel(ref.RBFKernel(sigma=0.5), dyadic_order=1)
    Xg = X.clone().requires_grad_(True)
    K = sk.compute_kernel(Xg, Y)
    K.sum().backward()
    grad_kernel = Xg.grad.clone()
    G = sk.compute_Gram(X, Y, sym=False)
    Xg = X.clone().requires_grad_(True)
    mmd = sk.compute_mmd(Xg, Y)
    mmd.backward()
    grad_mmd = Xg.grad.clone()
    dist = sk.compute_distance(X, Y)
    sr = sk.compute_scoring_rule(X, Z[:1])
    esr = sk.compute_expected_scoring_rule(X, Z)
    save("readme_c1", X=X, Y=Y, Z=Z, sigma=0.5, dyadic=1, kernel=K, grad_kernel_sum=grad_kernel, gram=G,
         mmd=mmd, grad_mmd=grad_mmd, distance=dist, scoring_rule=sr, expected_scoring_rule=esr)


# ---------------------------------------------------------------------------
# 3. Gram forward + adjoint on reduced C2/C3/C4-like inputs
# ---------------------------------------------------------------------------
def gram_cases():
    gen = torch.Generator().manual_seed(0)
    cases = [
        # name, kernel, param, dyadic, A, B, M, N, D, naive
        ("c2mini_rbf_d1", "rbf", 1.0, 1, 6, 6, 16, 16, 3, 0),
        ("c3mini_lin_d1", "linear", 0.0, 1, 5, 7, 24, 24, 8, 0),
        ("c4mini_rbf_d2", "rbf", 1.0, 2, 4, 5, 12, 12, 4, 0),
        ("lin_d0_ragged", "linear", 0.0, 0, 3, 4, 9, 14, 2, 0),
        ("rbf_d0_ragged", "rbf", 0.5, 0, 4, 3, 13, 7, 3, 0),
        ("lin_d2_ragged", "linear", 0.0, 2, 3, 3, 6, 11, 5, 0),
        ("rbf_d1_naive", "rbf", 1.0, 1, 3, 4, 8, 10, 2, 1),
        ("lin_d3", "linear", 0.0, 3, 2, 3, 5, 6, 3, 0),
        ("len2", "linear", 0.0, 1, 3, 2, 2, 2, 2, 0),
    ]
    for name, kn, param, d, A, B, M, N, D, naive in cases:
        X = walk(gen, A, M, D)
        Y = walk(gen, B, N, D)
        if kn == "rbf":  # rougher inputs so that increments are not tiny
            X = X * 2.0
            Y = Y * 2.0
        sk = ref.SigKernel(kernel_of(kn, param), dyadic_order=d, _naive_solver=bool(naive))
        gram = sk.compute_Gram(X, Y, sym=False)
        w = torch.randn(A, B, generator=gen, dtype=torch.float64)
        Xg = X.clone().requires_grad_(True)
        (sk.compute_Gram(Xg, Y, sym=False) * w).sum().backward()
        grad_w = Xg.grad.clone()
        # tiling equivalence (sigkernel.py:102-127): max_batch below the batch sizes
        gram_tiled = sk.compute_Gram(X, Y, sym=False, max_batch=2)
        out = dict(X=X, Y=Y, kernel=kn, param=param, dyadic=d, naive=naive, gram=gram, w=w, grad_w=grad_w,
                   gram_tiled=gram_tiled)
        if M == N:
            # X against itself: exercises sym=True and the 2x rule (sigkernel.py:410-412)
            Xg = X.clone().requires_grad_(True)
            Gs = sk.compute_Gram(Xg, Xg, sym=True)
            Gs.sum().backward()
            out.update(gram_xx_sym=Gs.detach(), grad_xx_sum=Xg.grad.clone())
            Xg = X.clone().requires_grad_(True)
            mmd = sk.compute_mmd(Xg, Y)
            mmd.backward()
            out.update(mmd=mmd.detach(), grad_mmd=Xg.grad.clone())
        if A == B or True:
            n = min(A, B)
            Xg = X[:n].clone().requires_grad_(True)
            Kp = sk.compute_kernel(Xg, Y[:n])
            wp = torch.randn(n, generator=gen, dtype=torch.float64)
            (Kp * wp).sum().backward()
            out.update(paired=Kp.detach(), wp=wp, grad_paired=Xg.grad.clone())
        save("gram_" + name, **out)


# ---------------------------------------------------------------------------
# 4. known-answer test: two straight lines, <dx,dy> = c  =>  k = I0(2 sqrt(c))
# ---------------------------------------------------------------------------
def kat_case():
    t = torch.linspace(0, 1, 2, dtype=torch.float64)[None, :, None]
    X = t.clone()
    Y = t.clone()
    vals = {}
    for d in (0, 1, 4, 8):
        sk = ref.SigKernel(ref.LinearKernel(), dyadic_order=d)
        vals["d%d" % d] = sk.compute_kernel(X, Y)
    save("kat_straight_lines", X=X, Y=Y, **vals)


# ---------------------------------------------------------------------------------------------
# 5. callers either side of the path: path transforms, hypothesis test statistic, SigCHSIC
# ---------------------------------------------------------------------------------------------
def wrappers_case():
    import contextlib, io
    gen = torch.Generator().manual_seed(5)
    P = walk(gen, 4, 9, 2)
    out = dict(paths=P)
    for at in (0, 1):
        for ll in (0, 1):
            out["transform_at%d_ll%d" % (at, ll)] = ref.transform(P.numpy(), at=bool(at), ll=bool(ll), scale=0.5)
    X, Y, Z = walk(gen, 6, 8, 2) * 2, walk(gen, 6, 8, 2) * 2, walk(gen, 6, 7, 3) * 2
    k = ref.RBFKernel(1.0)
    out.update(X=X, Y=Y, Z=Z, chsic=ref.SigCHSIC(X, Y, Z, k, dyadic_order=1, eps=0.1))
    sk = ref.SigKernel(k, 0)
    out.update(mmd_d0=sk.compute_mmd(X, Y), c_alpha=ref.c_alpha(6, 0.99))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref.hypothesis_test(X, Y, k, confidence_level=0.99, dyadic_order=0)
    out["verdict_rejected"] = int("rejected" in buf.getvalue())
    save("wrappers", **out)


if __name__ == "__main__":
    solver_cases()
    readme_case()
    gram_cases()
    kat_case()
    wrappers_case()
