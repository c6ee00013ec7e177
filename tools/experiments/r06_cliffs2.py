#!/usr/bin/env python3
"""More neighbouring-size timings: the derivative Gram, long fp32 paths (C5's regime), the loss wrappers by batch size.  usage: r06_cliffs2.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
g = torch.Generator().manual_seed(0)
def walk(A, M, D, dt=torch.float64): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()
def t(f, n=3, reps=3):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2]
RBF, LIN = sigkernel_amd.RBFKernel, sigkernel_amd.LinearKernel
print("== derivative Gram (rbf dim 8, d=1): batch, then length")
for A, M in ((63, 128), (64, 128), (65, 128), (127, 128), (128, 128), (129, 128), (255, 128), (256, 128), (257, 128), (128, 125), (128, 126), (128, 127), (128, 129), (128, 157), (128, 158), (128, 200)):
    sk = sigkernel_amd.SigKernel(RBF(1.0), 1); X, Y, G = walk(A, M, 8), walk(A, M, 8), torch.randn(A, M, 8, generator=g, dtype=torch.float64).cuda()
    ms = t(lambda: sk.compute_kernel_and_derivatives_Gram(X, Y, G))
    print("%4d x %-4d pairs of %3d points | %8.3f ms %8.1f ns/pair" % (A, A, M, ms, ms * 1e6 / (A * A)), flush=True)
print("== long fp32 paths, rbf dim 16, d=2 (C5's regime), 64 x 64 pairs: length")
for M in (255, 256, 257, 300, 511, 512, 513, 520, 600):
    sk = sigkernel_amd.SigKernel(RBF(1.0), 2); X, Y = walk(64, M, 16, torch.float32), walk(64, M, 16, torch.float32)
    ms = t(lambda: sk.compute_Gram(X, Y))
    def step():
        Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
    ms2 = t(step)
    print("%3d points | forward %8.3f ms %6.2f ps/cell | forward + backward %8.3f ms" % (M, ms, ms * 1e9 / (64 * 64 * ((M - 1) * 4) ** 2), ms2), flush=True)
print("== loss wrappers (rbf dim 3, 64 points, d=1): compute_mmd + backward by batch")
for A in (31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 400, 511, 512, 513, 560, 561, 562, 700):
    sk = sigkernel_amd.SigKernel(RBF(1.0), 1); X, Y = walk(A, 64, 3), walk(A, 64, 3)
    def step():
        Xg = X.clone().requires_grad_(True); sk.compute_mmd(Xg, Y).backward()
    ms = t(step)
    print("%4d + %-4d paths | %8.3f ms %7.1f ns/pair" % (A, A, ms, ms * 1e6 / (3 * A * A)), flush=True)
