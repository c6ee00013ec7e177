#!/usr/bin/env python3
"""Randomised differential test of the whole public API on the GPU against the oracle (the reference's CPU algorithm, oracle/).

Every case draws a static kernel (LinearKernel with a scale, RBFKernel with a sigma, or a user-defined duck-typed kernel that takes
the generic route), a dyadic order 0..4, either stencil, fp64 or fp32, batch sizes 1..24, path lengths 2..90 (one case in three with
equal lengths: the merged loss route; one in seven 100..420 points: several bands per pair), path dimension 1..20 (one case in four 17..36), the default transient budget or a tiny one (every call tiles over rows), the default routes or memory-first
(routes.no_stream), and checks against the oracle's closed forms
  compute_kernel        values + gradient under random weights            (_SigKernel, sigkernel.py:201-343)
  compute_Gram          values (sym or not) + gradient, 2x rule            (_SigKernelGram, :347-416; prep_backward :419-502)
  compute_mmd, compute_scoring_rule, compute_expected_scoring_rule, compute_distance: values + gradient (:130-197)
  compute_kernel_and_derivatives_Gram                                      (k_kgrad, :504-593)
usage: fuzz_api.py [n_cases] [seed] ; prints one line per failure and a summary; exit status 1 on any failure."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sigkernel_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402

DEV = "cuda"


class CauchyKernel:
    """A user-defined static kernel (not one of the two the fused kernels implement): k(x, y) = 1 / (1 + |x - y|^2 / c)."""

    def __init__(self, c):
        self.c = c

    def batch_kernel(self, X, Y):
        d2 = ((X[:, :, None, :] - Y[:, None, :, :]) ** 2).sum(-1)
        return 1.0 / (1.0 + d2 / self.c)

    def Gram_matrix(self, X, Y):
        d2 = ((X[:, None, :, None, :] - Y[None, :, None, :, :]) ** 2).sum(-1)
        return 1.0 / (1.0 + d2 / self.c)


class _Paired:
    """The oracle's Gram route on ONE pair with batch_kernel's scaling (LinearKernel: static_kernels.py:24 against :33)."""

    def __init__(self, k):
        self.k = k

    def Gram_matrix(self, X, Y):
        return self.k.batch_kernel(X, Y)[:, None]


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))) if a.size else 0.0


def walk(rng, A, M, D):
    return torch.from_numpy(np.cumsum(rng.standard_normal((A, M, D)), axis=1) * (0.6 / np.sqrt(M * D)))


def draw(rng):
    c = {}
    c["kind"] = str(rng.choice(["linear", "rbf", "rbf", "cauchy"]))
    c["param"] = float(rng.uniform(0.5, 1.5))
    c["dyadic"] = int(rng.choice([0, 1, 1, 2, 2, 3, 4]))
    c["naive"] = bool(rng.integers(0, 2))
    c["f32"] = bool(rng.integers(0, 4) == 0)
    c["A"], c["B"] = int(rng.integers(1, 25)), int(rng.integers(1, 25))
    c["M"] = int(rng.integers(2, 91))
    c["N"] = c["M"] if rng.integers(0, 3) == 0 else int(rng.integers(2, 91))
    c["D"] = int(rng.integers(1, 21)) if rng.integers(0, 4) else int(rng.integers(17, 37))      # (one in four: 17..36 dims, the widest static kernels and beyond)
    if rng.integers(0, 7) == 0:    # long paths: several bands per pair (the multi-band kernels), few of them
        c["M"], c["N"] = int(rng.integers(100, 420)), int(rng.integers(100, 420))
        c["A"], c["B"] = min(c["A"], 4), min(c["B"], 4)
    if c["dyadic"] >= 3:
        c["M"], c["N"] = min(c["M"], 30 if c["dyadic"] == 3 else 16), min(c["N"], 30 if c["dyadic"] == 3 else 16)
    c["workspace"] = int(rng.choice([0, 0, 1 << 16, 1 << 20]))      # 0: the default budget; small: every call tiles over rows
    c["memory_first"] = bool(rng.integers(0, 3) == 0)                # routes.no_stream
    if c["kind"] == "cauchy":      # the generic route builds (A, B, M, N, D) differences in torch: keep it small
        c["A"], c["B"], c["M"], c["N"], c["D"] = min(c["A"], 6), min(c["B"], 6), min(c["M"], 30), min(c["N"], 30), min(c["D"], 6)
    return c


def make_kernel(c):
    if c["kind"] == "linear":
        return sigkernel_amd.LinearKernel(c["param"])
    if c["kind"] == "rbf":
        return sigkernel_amd.RBFKernel(c["param"])
    return CauchyKernel(c["param"])


def run_case(c, rng):
    """Returns a list of (what, error, tolerance) that failed."""
    k, d, nv = make_kernel(c), c["dyadic"], c["naive"]
    dt = torch.float32 if c["f32"] else torch.float64
    ftol, gtol = (3e-4, 3e-3) if c["f32"] else (1e-10, 1e-8)
    A, B, M, N, D = c["A"], c["B"], c["M"], c["N"], c["D"]
    X, Y = walk(rng, A, M, D).to(dt), walk(rng, B, N, D).to(dt)
    Xo, Yo = X.double(), Y.double()          # the oracle sees exactly the values the device gets
    sk = sigkernel_amd.SigKernel(k, d, _naive_solver=nv, workspace_bytes=c.get("workspace") or None)
    sigkernel_amd.routes.no_stream = bool(c.get("memory_first"))
    bad = []

    def check(what, got, want, tol):
        e = rel_err(got, want)
        if not e <= tol:
            bad.append((what, e, tol))

    def dev(t, grad=False):
        t = t.to(DEV)
        return t.requires_grad_(True) if grad else t
    # ---- Gram, values and gradient under random weights
    Kw = O.gram_forward(Xo, Yo, k, d, naive=nv)
    w = rng.standard_normal((A, B))
    Xg = dev(X, True)
    K = sk.compute_Gram(Xg, dev(Y))
    (K * torch.from_numpy(w).to(dt).to(DEV)).sum().backward()
    check("gram", K.detach().cpu().numpy(), Kw, ftol)
    check("gram no-grad", sk.compute_Gram(dev(X), dev(Y)).cpu().numpy(), Kw, ftol)
    check("gram grad", Xg.grad.cpu().numpy(), O.gram_grad_weighted(Xo, Yo, w, k, d, naive=nv), gtol)
    # ---- symmetric Gram with a (non-symmetric) upstream gradient: the reference's 2x rule on the first-argument gradient
    Kxx = O.gram_forward(Xo, Xo, k, d, naive=nv)
    ws = rng.standard_normal((A, A))
    Xg = dev(X, True)
    Ks = sk.compute_Gram(Xg, Xg, sym=True)
    (Ks * torch.from_numpy(ws).to(dt).to(DEV)).sum().backward()
    check("gram sym", Ks.detach().cpu().numpy(), Kxx, ftol)
    if not torch.equal(Ks, Ks.t()):
        bad.append(("gram sym: not exactly symmetric", 1.0, 0.0))
    # (the triangular routes fold the mirror pairs through the second argument: exact for ANY upstream gradient)
    gs = O.gram_grad_weighted(Xo, Xo, ws, k, d, naive=nv) + O.gram_grad_weighted(Xo, Xo, ws.T.copy(), k, d, naive=nv)
    gref = 2.0 * O.gram_grad_weighted(Xo, Xo, ws, k, d, naive=nv)
    got = Xg.grad.cpu().numpy()
    if min(rel_err(got, gs), rel_err(got, gref)) > gtol:      # 2x rule (all pairs) or first + second argument (triangle): equal for symmetric ws
        bad.append(("gram sym grad", min(rel_err(got, gs), rel_err(got, gref)), gtol))
    # ---- paired
    n = min(A, B)
    if M >= 2 and N >= 2:
        G = k.batch_kernel(Xo[:n], Yo[:n]).numpy()
        kw = O.solve_coarse(O.increments(G), d, nv)
        v = rng.standard_normal(n)
        Xg = dev(X[:n], True)
        kk = sk.compute_kernel(Xg, dev(Y[:n]))
        (kk * torch.from_numpy(v).to(dt).to(DEV)).sum().backward()
        check("kernel", kk.detach().cpu().numpy(), kw, ftol)
        gp = np.stack([O.gram_grad_weighted(Xo[i:i + 1], Yo[i:i + 1], np.array([[v[i]]]), _Paired(k), d, naive=nv)[0] for i in range(n)])
        check("kernel grad", Xg.grad.cpu().numpy(), gp, gtol)
        if M == N:      # (the gradient of the distance: first arguments of k(x_i, x_i) and k(x_i, y_i) only, as the reference's _SigKernel)
            Xg = dev(X[:n], True)
            sk.compute_distance(Xg, dev(Y[:n])).backward()
            gd = np.stack([O.gram_grad_weighted(Xo[i:i + 1], Xo[i:i + 1], np.array([[1.0 / n]]), _Paired(k), d, naive=nv)[0]
                           + O.gram_grad_weighted(Xo[i:i + 1], Yo[i:i + 1], np.array([[-2.0 / n]]), _Paired(k), d, naive=nv)[0] for i in range(n)])
            check("distance grad", Xg.grad.cpu().numpy(), gd, gtol)
        check("distance", float(sk.compute_distance(dev(X[:n]), dev(Y[:n]))),
              O.solve_coarse(O.increments(k.batch_kernel(Xo[:n], Xo[:n]).numpy()), d, nv).mean()
              + O.solve_coarse(O.increments(k.batch_kernel(Yo[:n], Yo[:n]).numpy()), d, nv).mean() - 2.0 * kw.mean(), 50 * ftol)
    # ---- loss wrappers
    if A >= 2:
        Kyy = O.gram_forward(Yo, Yo, k, d, naive=nv)
        wxx = (1.0 - np.eye(A)) / (A * (A - 1.0))
        for name, Bv in (("mmd", B), ("expected_scoring_rule", B), ("scoring_rule", 1)):
            if name == "mmd" and B < 2:
                continue
            Yv, Yvo = Y[:Bv], Yo[:Bv]
            Kxy = Kw[:, :Bv]
            want = float((Kxx * wxx).sum() - 2.0 * Kxy.mean())
            if name == "mmd":
                want += float((Kyy.sum() - np.trace(Kyy)) / (B * (B - 1.0)))
            Xg = dev(X, True)
            val = getattr(sk, "compute_" + name)(Xg, dev(Yv))
            val.backward()
            val = val.detach()
            scale = max(1.0, abs(want), float(np.abs(Kxx).max()))
            if not abs(float(val) - want) <= 20 * ftol * scale:
                bad.append((name, abs(float(val) - want) / scale, 20 * ftol))
            gw = 2.0 * O.gram_grad_weighted(Xo, Xo, wxx, k, d, naive=nv) + O.gram_grad_weighted(Xo, Yvo, np.full((A, Bv), -2.0 / (A * Bv)), k, d, naive=nv)
            check(name + " grad", Xg.grad.cpu().numpy(), gw, gtol)
    # ---- kernel and its two directional derivatives
    if not c["f32"]:
        gam = walk(rng, A, M, D)
        want3 = O.kgrad(Xo, Yo, gam, k, d)
        got3 = sk.compute_kernel_and_derivatives_Gram(dev(X), dev(Y), dev(gam))
        for nm, g3, w3, tol in zip(("k", "k'", "k''"), got3, want3, (1e-10, 1e-6, 1e-3)):
            check("kgrad " + nm, g3.cpu().numpy(), w3, tol)
    sigkernel_amd.routes.no_stream = False
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    fails = 0
    import time
    t0, slow = time.time(), float(os.environ.get("FUZZ_SLOW", "0") or 0)      # FUZZ_SLOW=s: name the cases that take longer than s seconds
    for i in range(n):
        c = draw(rng)
        t1 = time.time()
        if slow:
            print("case %d %s" % (i, c), file=sys.stderr, flush=True)
        try:
            bad = run_case(c, rng)
        except Exception:      # noqa: BLE001
            bad = [("exception", 1.0, 0.0)]
            traceback.print_exc()
        if slow and time.time() - t1 > slow:
            print("SLOW case %d: %.1f s (%.1f s so far) %s" % (i, time.time() - t1, time.time() - t0, c), flush=True)
        if bad:
            fails += 1
            print("FAIL case %d %s: %s" % (i, c, "; ".join("%s %.2e > %.0e" % b for b in bad)), flush=True)
    print("fuzz_api: %d cases, seed %d, %d failed" % (n, seed, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
