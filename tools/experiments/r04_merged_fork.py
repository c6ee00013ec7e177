#!/usr/bin/env python3
"""The merged loss route's K_YY on a side stream while a hipGraph is captured (parallel branch) against everything on one stream."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
for A, M, D, d, kern in ((16, 64, 3, 1, "rbf"), (32, 64, 3, 1, "rbf"), (64, 64, 3, 1, "rbf"), (128, 64, 3, 1, "rbf"), (64, 128, 8, 1, "linear"), (32, 32, 4, 2, "rbf")):
    X, Y = walk(A, M, D), walk(A, M, D)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0) if kern == "rbf" else sigkernel_amd.LinearKernel(), d)
    def step(Xg): sk.compute_mmd(Xg, Y).backward()
    res = {}
    for rnd in range(3):
        for one in (True, False):
            sigkernel_amd.routes.no_mmd_streams = one
            sX = X.clone().requires_grad_(True)
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3): step(sX); sX.grad = None
            torch.cuda.current_stream().wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr): step(sX)
            for _ in range(3): gr.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): gr.replay()
            torch.cuda.synchronize()
            res.setdefault(one, []).append((time.perf_counter() - t0) / 100 * 1e3)
            del gr
    print("%-6s A=B=%3d len %3d dim %d d=%d | hipGraph replay: one stream %.3f ms, K_YY forked %.3f ms" % (kern, A, M, D, d, min(res[True]), min(res[False])), flush=True)
