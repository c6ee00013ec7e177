#!/usr/bin/env python3
"""Cliff finder: per-pair cost of the public calls at NEIGHBOURING sizes (batch 127 / 128 / 129 ..., lengths 63 / 64 / 65 ...): a jump
between neighbours that the work does not explain is a scheduling or routing cliff (the one found this way: the fused adjoints' chunk
length had to divide the batch, profiles/r06_batch_divisors.txt).  usage: r06_cliffs.py -> profiles/r06_cliffs.txt"""
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
def ops(sk, X, Y):
    A, B = X.shape[0], Y.shape[0]
    w = torch.randn(A, B, generator=g, dtype=torch.float64).to(X.dtype).cuda()
    def gram(): sk.compute_Gram(X, Y)
    def gram_b():
        Xg = X.clone().requires_grad_(True); (sk.compute_Gram(Xg, Y) * w).sum().backward()
    def sym_b():
        Xg = X.clone().requires_grad_(True); (sk.compute_Gram(Xg, Xg, sym=True) * w[:, :A]).sum().backward()
    def mmd_b():
        Xg = X.clone().requires_grad_(True); sk.compute_mmd(Xg, Y).backward()
    def pair_b():
        Xg = X.clone().requires_grad_(True); sk.compute_kernel(Xg, Y).sum().backward()
    return (("Gram", gram, A * B), ("Gram+bwd", gram_b, A * B), ("symGram+bwd", sym_b, A * A), ("mmd+bwd", mmd_b, 2 * A * A + A * B), ("paired+bwd", pair_b, A))
def sweep(title, cases):
    print("== " + title)
    prev = {}
    for label, kind, A, M, D, d, dt in cases:
        k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
        sk = sigkernel_amd.SigKernel(k, d)
        X, Y = walk(A, M, D, dt), walk(A, M, D, dt)
        row = []
        for name, f, pairs in ops(sk, X, Y):
            ms = t(f)
            ns = ms * 1e6 / pairs
            flag = ""
            if name in prev and (ns > 1.5 * prev[name] or ns < prev[name] / 1.5) and name != "paired+bwd": flag = " <<"
            prev[name] = ns
            row.append("%s %8.3f ms %8.1f ns/pair%s" % (name, ms, ns, flag))
        print("%-34s | %s" % (label, " | ".join(row)), flush=True)
f64, f32 = torch.float64, torch.float32
for kind in ("linear", "rbf"):
    D = 8 if kind == "linear" else 3
    sweep("%s, batch size (64 points, dim %d, d=1, fp64)" % (kind, D), [("%d paths" % A, kind, A, 64, D, 1, f64) for A in (63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 383, 511, 512, 513)])
    sweep("%s, path length (128 paths, dim %d, d=1, fp64)" % (kind, D), [("%d points" % M, kind, 128, M, D, 1, f64) for M in (31, 32, 33, 63, 64, 65, 66, 127, 128, 129, 130, 255, 257)])
    sweep("%s, path dim (128 paths of 64 points, d=1, fp64)" % kind, [("dim %d" % Dd, kind, 128, 64, Dd, 1, f64) for Dd in (1, 4, 5, 8, 9, 16, 17)])
    sweep("%s, dyadic order / dtype (128 paths of 64 points, dim %d)" % (kind, D), [("d=%d %s" % (d, "fp32" if dt == f32 else "fp64"), kind, 128, 64, D, d, dt) for d in (0, 1, 2, 3) for dt in (f64, f32)])
