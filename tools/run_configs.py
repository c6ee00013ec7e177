#!/usr/bin/env python3
"""Run BASELINE.json configs[0..4] at full size on one MI355X: timing + spot-check parity against the CPU oracle.
usage: python tools/run_configs.py [c1 c2 c3 c4 c5 ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sigkernel_amd
from oracle import oracle as O

dev = "cuda:0"


def walk(gen, A, M, D, dtype=torch.float64):
    return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dtype)


def spot(K, Xc, Yc, kern, d, n=16, seed=0):
    rng = np.random.default_rng(seed)
    A, B = K.shape
    worst = 0.0
    for p in rng.integers(0, A * B, size=n):
        a, b = divmod(int(p), B)
        want = O.gram_forward(Xc[a:a + 1].double(), Yc[b:b + 1].double(), kern, d)[0, 0]
        worst = max(worst, abs(float(K[a, b]) - want) / abs(want))
    return worst


def sync_time(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, time.perf_counter() - t0


which = sys.argv[1:] or ["c1", "c2", "c3", "c4", "c5", "c6"]
gen = torch.Generator().manual_seed(0)
if "c1" in which:
    torch.manual_seed(0)
    X, Y = torch.rand(5, 10, 2, dtype=torch.float64), torch.rand(5, 20, 2, dtype=torch.float64)
    k = sigkernel_amd.RBFKernel(0.5); sk = sigkernel_amd.SigKernel(k, 1)
    Xg = X.to(dev).requires_grad_(True)
    K = sk.compute_Gram(Xg, Y.to(dev)); mmd = sk.compute_mmd(Xg, Y.to(dev)); mmd.backward()
    print("c1  README 5x5 len 10/20 RBF d=1: max rel err vs oracle %.1e, mmd %.6f, |grad|max %.4f"
          % (spot(K.detach().cpu(), X, Y, k, 1, 25), float(mmd.detach()), float(Xg.grad.abs().max())))
if "c2" in which:
    X = walk(gen, 128, 64, 3); k = sigkernel_amd.RBFKernel(1.0); sk = sigkernel_amd.SigKernel(k, 1)
    Xd = X.to(dev); sk.compute_Gram(Xd, Xd, sym=True)
    K, t = sync_time(lambda: sk.compute_Gram(Xd, Xd, sym=True))
    print("c2  128x128 len 64 dim 3 RBF d=1 sym: %.2f ms, %.3e entries/s, max rel err %.1e, |K-K^T|max %.1e"
          % (t * 1e3, 128 * 128 / t, spot(K.cpu(), X, X, k, 1), float((K - K.t()).abs().max())))
if "c3" in which:
    X, Y = walk(gen, 512, 128, 8), walk(gen, 512, 128, 8); k = sigkernel_amd.LinearKernel(); sk = sigkernel_amd.SigKernel(k, 1)
    Xd, Yd = X.to(dev), Y.to(dev); sk.compute_Gram(Xd, Yd)
    K, t = sync_time(lambda: sk.compute_Gram(Xd, Yd))
    print("c3  512x512 len 128 dim 8 Linear d=1: %.2f ms, %.3e entries/s, %.3e cells/s, max rel err %.1e"
          % (t * 1e3, 512 * 512 / t, 512 * 512 * 254 * 254 / t, spot(K.cpu(), X, Y, k, 1)))
if "c4" in which:
    A = int(os.environ.get("C4_BATCH", "2048"))
    X, Y = walk(gen, A, 64, 4), walk(gen, A, 64, 4); k = sigkernel_amd.RBFKernel(1.0); sk = sigkernel_amd.SigKernel(k, 2)
    Xd, Yd = X.to(dev), Y.to(dev); sk.compute_Gram(Xd, Yd)
    K, t1 = sync_time(lambda: sk.compute_Gram(Xd, Yd))
    Xg = Xd.clone().requires_grad_(True)
    mmd, tf = sync_time(lambda: sk.compute_mmd(Xg, Yd))
    _, tb = sync_time(lambda: mmd.backward())
    cells = A * A * 252 * 252
    print("c4  %dx%d len 64 dim 4 RBF d=2 (1 GPU): Gram %.1f ms (%.3e entries/s, %.3e cells/s), max rel err %.1e; "
          "mmd fwd %.1f ms + bwd %.1f ms, mmd %.3e, peak mem %.1f GB"
          % (A, A, t1 * 1e3, A * A / t1, cells / t1, spot(K.cpu(), X, Y, k, 2, 8), tf * 1e3, tb * 1e3, float(mmd.detach()),
             torch.cuda.max_memory_allocated() / 1e9))
    # gradient spot check: 2 rows against the oracle's closed form (restricted to 8 columns of Y to stay cheap)
    Xs, Ys = X[:2], Y[:8]
    Xg2 = Xs.to(dev).requires_grad_(True)
    w = torch.linspace(-1, 1, 16, dtype=torch.float64).reshape(2, 8)
    (sk.compute_Gram(Xg2, Ys.to(dev)) * w.to(dev)).sum().backward()
    gp = O.gram_grad_points(Xs, Ys, k, 2, nthreads=8)
    want = np.einsum("ab,abmd->amd", w.numpy(), gp)
    print("    adjoint spot check: max-norm rel err %.1e" % (np.abs(Xg2.grad.cpu().numpy() - want).max() / np.abs(want).max()))
if "c5" in which:
    X, Y = walk(gen, 256, 512, 16, torch.float32), walk(gen, 256, 512, 16, torch.float32)
    k = sigkernel_amd.RBFKernel(1.0); sk = sigkernel_amd.SigKernel(k, 2)
    Xd, Yd = X.to(dev), Y.to(dev); sk.compute_Gram(Xd, Yd)
    K, t = sync_time(lambda: sk.compute_Gram(Xd, Yd))
    print("c5  256x256 len 512 dim 16 RBF d=2 fp32 (grid 2044^2): %.1f ms, %.3e entries/s, %.3e cells/s, "
          "max rel err vs fp64 oracle %.1e (4 pairs)" % (t * 1e3, 256 * 256 / t, 256 * 256 * 2044 * 2044 / t, spot(K.cpu(), X, Y, k, 2, 4)))
if "c6" in which:
    # not a BASELINE config: gradients on grids beyond the reference's 1024-thread limit (fused adjoint, multi-band strips)
    X, Y = walk(gen, 64, 700, 4), walk(gen, 64, 700, 4)
    k = sigkernel_amd.RBFKernel(1.0); sk = sigkernel_amd.SigKernel(k, 1)
    Xd, Yd = X.to(dev), Y.to(dev)
    Xg = Xd.clone().requires_grad_(True)
    sk.compute_mmd(Xg, Yd).backward()
    Xg = Xd.clone().requires_grad_(True)
    mmd, tf = sync_time(lambda: sk.compute_mmd(Xg, Yd))
    _, tb = sync_time(lambda: mmd.backward())
    Xs, Ys = X[:2], Y[:3]
    Xg2 = Xs.to(dev).requires_grad_(True)
    w = torch.linspace(-1, 1, 6, dtype=torch.float64).reshape(2, 3)
    (sk.compute_Gram(Xg2, Ys.to(dev)) * w.to(dev)).sum().backward()
    gp = O.gram_grad_points(Xs, Ys, k, 1, nthreads=8)
    want = np.einsum("ab,abmd->amd", w.numpy(), gp)
    print("c6  64x64 len 700 dim 4 RBF d=1 (grid 1398^2, beyond the reference's CUDA limit): mmd fwd %.1f ms + bwd %.1f ms; "
          "gradient max-norm rel err vs oracle %.1e" % (tf * 1e3, tb * 1e3, np.abs(Xg2.grad.cpu().numpy() - want).max() / np.abs(want).max()))
