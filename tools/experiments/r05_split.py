#!/usr/bin/env python3
"""Few pairs of long paths: the multi-band forward with the bands of a pair on several waves (split mode, SK_FUSEDMB_SPLIT) against the
one-wave-per-pair sweep -- time and bits.  usage: r05_split.py [quick]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
lib = _lib.load()
g = torch.Generator().manual_seed(0)
def walk(A, M, D, dt=torch.float64): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()
def run(sk, X, Y, split, reps):
    os.environ["SK_FUSEDMB_SPLIT"] = "1" if split else "0"
    lib.sk_reload_knobs()
    K = sk.compute_Gram(X, Y); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): K = sk.compute_Gram(X, Y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, K
cases = [("linear", 16, 16, 4096, 4, 0, torch.float64), ("linear", 1, 1, 8192, 4, 0, torch.float64), ("rbf", 16, 16, 4096, 4, 0, torch.float64), ("rbf", 1, 1, 8192, 4, 0, torch.float64), ("rbf", 16, 16, 4096, 4, 1, torch.float64),
         ("linear", 16, 16, 4096, 8, 1, torch.float64), ("rbf", 1, 1, 8192, 3, 2, torch.float64), ("rbf", 4, 4, 2048, 16, 2, torch.float32),
         ("linear", 2, 3, 1500, 12, 0, torch.float64), ("rbf", 8, 8, 1000, 5, 1, torch.float64), ("rbf", 32, 32, 700, 3, 1, torch.float64),
         ("rbf", 2, 2, 2048, 3, 1, torch.float64), ("rbf", 1, 7, 4095, 2, 0, torch.float64)]
if len(sys.argv) > 1: cases = cases[:4]
for kind, A, B, M, D, d, dt in cases:
    X, Y = walk(A, M, D, dt), walk(B, M + 3, D, dt)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0) if kind == "rbf" else sigkernel_amd.LinearKernel(), d)
    reps = 10
    t1, K1 = run(sk, X, Y, False, reps)
    t2, K2 = run(sk, X, Y, True, reps)
    t2b, K2b = run(sk, X, Y, True, reps)
    print("%-6s %2dx%-2d len %4d dim %2d d=%d %s | one wave per pair %8.2f ms | bands on several waves %8.2f ms (%.1fx) | same bits %s, rerun %s | K[0,0]=%.12g"
          % (kind, A, B, M, D, d, "f64" if dt == torch.float64 else "f32", t1, t2, t1 / t2, bool(torch.equal(K1, K2)), bool(torch.equal(K2, K2b)), float(K2[0, 0])), flush=True)
