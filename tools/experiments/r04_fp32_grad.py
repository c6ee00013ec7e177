#!/usr/bin/env python3
"""fp32 paths (torch's default dtype: what ML users pass) with a gradient, against the same paths in fp64."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (A, M, D, d, kern) in ((64, 64, 3, 1, "rbf"), (512, 64, 4, 2, "rbf"), (512, 128, 8, 1, "linear"), (128, 300, 6, 0, "rbf"), (256, 32, 4, 0, "rbf")):
    X64, Y64 = walk(A, M, D), walk(A, M, D)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0) if kern == "rbf" else sigkernel_amd.LinearKernel(), d)
    out = {}
    for dt in (torch.float64, torch.float32):
        X, Y = X64.to(dt), Y64.to(dt)
        w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda().to(dt)
        Xg = X.clone().requires_grad_(True)
        def gram():
            Xg.grad = None; (sk.compute_Gram(Xg, Y) * w).sum().backward()
        def mmd():
            Xg.grad = None; sk.compute_mmd(Xg, Y).backward()
        out[dt] = (t(lambda: sk.compute_Gram(X, Y)), t(gram), t(mmd))
    print("%-6s A=%3d len %3d dim %d d=%d | forward fp64 %.3f fp32 %.3f | Gram + backward fp64 %.3f fp32 %.3f | mmd + backward fp64 %.3f fp32 %.3f  (ms)" % (
        kern, A, M, D, d, out[torch.float64][0], out[torch.float32][0], out[torch.float64][1], out[torch.float32][1], out[torch.float64][2], out[torch.float32][2]), flush=True)
