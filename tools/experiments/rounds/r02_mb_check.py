#!/usr/bin/env python3
"""GPU box: the multi-band fused forward (sk_solve_fwd_static_*) against the streaming route on random shapes, then C5 timing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import _lib
from sigkernel_amd.sigkernel import _increments

be = _lib.get_backend()
dev = "cuda:0"


def walk(gen, A, M, D, dtype=torch.float64):
    return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)).to(dtype)


rng = np.random.default_rng(0)
bad = 0
n = 0
for it in range(int(os.environ.get("N_IT", "60"))):
    kind = int(rng.integers(0, 2))
    d = int(rng.integers(0, 3))
    D = int(rng.integers(1, 17))
    M = int(rng.integers(2, 700 >> d)) if it % 3 else int(rng.integers(60, 140))
    N = int(rng.integers(150, 420))
    A, B = int(rng.integers(1, 6)), int(rng.integers(1, 8))
    gram = bool(it % 4)
    if it % 7 == 6:
        A, B, M, N = 40, 90, int(rng.integers(20, 200 >> d) + 2), 150 + int(rng.integers(0, 30))    # many pairs per wave
    if not gram:
        B = A
    dt = torch.float32 if it % 5 == 4 else torch.float64
    gen = torch.Generator().manual_seed(100 + it)
    X, Y = (walk(gen, A, M, D, dt) * 1.5).to(dev), (walk(gen, B, N, D, dt) * 1.5).to(dev)
    sk = sigkernel_amd.LinearKernel(0.9) if kind == 0 else sigkernel_amd.RBFKernel(0.8)
    par = (1.0 if gram else 0.9) if kind == 0 else 0.8
    K = be.solve_fwd_fused_static(kind, par, X, Y, d, False, gram)
    if K is None:
        print("it", it, "unsupported", kind, d, D, M, N)
        continue
    inc = _increments(be, sk, X.double(), Y.double(), gram)
    want = be.solve_fwd(inc, d)
    err = float((K.double() - want).abs().max() / want.abs().max())
    tol = 1e-11 if dt == torch.float64 else 2e-6
    n += 1
    if not err <= tol:
        bad += 1
        print("MISMATCH it=%d kind=%d d=%d D=%d A=%d B=%d M=%d N=%d gram=%s dt=%s err=%.3e" % (it, kind, d, D, A, B, M, N, gram, dt, err))
print("checked %d shapes, %d bad" % (n, bad))

# ---- C5 timing
gen = torch.Generator().manual_seed(1)
X, Y = walk(gen, 256, 512, 16, torch.float32).to(dev), walk(gen, 256, 512, 16, torch.float32).to(dev)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2)
for wpc in os.environ.get("WPCS", "0").split(","):
    if wpc != "0":
        os.environ["SK_FUSEDMB_WPC"] = wpc
    for _ in range(2):
        K = sk.compute_Gram(X, Y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        K = sk.compute_Gram(X, Y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("C5 compute_Gram wpc=%s: %.1f ms  -> %.3e cells/s" % (wpc, ms, 65536 * 2044.0 * 2044 / (ms * 1e-3)))
os.environ["SK_NO_FUSED_MB"] = "1"; sigkernel_amd.routes.reload()
K2 = sk.compute_Gram(X, Y)
print("C5 fused-mb vs streaming route: rel diff %.3e" % float((K - K2).abs().max() / K2.abs().max()))
# linear long paths, fp64
Xl, Yl = walk(gen, 128, 300, 8).to(dev), walk(gen, 128, 300, 8).to(dev)
skl = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1)
os.environ.pop("SK_NO_FUSED_MB"); sigkernel_amd.routes.reload()
for env in ("", "1"):
    if env:
        os.environ["SK_NO_FUSED_MB"] = "1"; sigkernel_amd.routes.reload()
    for _ in range(2):
        K = skl.compute_Gram(Xl, Yl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        K = skl.compute_Gram(Xl, Yl)
    torch.cuda.synchronize()
    print("linear 128x128 len 300 dim 8 d=1 (no_mb=%s): %.2f ms" % (env, (time.perf_counter() - t0) / 5 * 1e3))
