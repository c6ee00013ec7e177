#!/usr/bin/env python3
"""compute_mmd(X, Y).backward() through the merged route (sigkernel._SigKernelLoss: one Gram block K(X, [X; Y])) against the
reference's composition of three compute_Gram calls (routes.no_merged_loss), eager and replayed from a hipGraph; same box,
alternating.  Sizes above the route's own limit are forced through it (to place the limit)."""
import os, sys, time, faulthandler
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import sigkernel as S
S._SYM_MIN_CELLS = 1e30     # (also keeps the composition's K_XX off the blocked triangle: compare tools/ab.py c4 for that)
g = torch.Generator().manual_seed(0)
def walk(A, M, D, dt=torch.float64): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()
CASES = ((16, 64, 3, 1, "rbf"), (32, 64, 3, 1, "rbf"), (64, 64, 3, 1, "rbf"), (128, 64, 3, 1, "rbf"), (192, 64, 3, 1, "rbf"), (256, 64, 3, 1, "rbf"), (512, 64, 3, 1, "rbf"),
         (64, 128, 8, 1, "linear"), (128, 128, 8, 1, "linear"), (256, 128, 8, 1, "linear"), (32, 32, 4, 2, "rbf"), (128, 64, 4, 2, "rbf"), (256, 64, 4, 2, "rbf"),
         (64, 200, 7, 0, "rbf"), (32, 512, 16, 2, "rbf32"))
if len(sys.argv) > 1:
    CASES = [CASES[int(i)] for i in sys.argv[1].split(',')]
for A, M, D, d, kern in CASES:
    dt = torch.float32 if kern == "rbf32" else torch.float64
    X, Y = walk(A, M, D, dt), walk(A, M, D, dt)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel() if kern == "linear" else sigkernel_amd.RBFKernel(1.0), d)
    def step(Xg):
        v = sk.compute_mmd(Xg, Y)
        v.backward()
        return v.detach()
    res, grads, vals = {}, {}, {}
    n_e, n_g = (30, 50) if A <= 128 else (10, 10)
    for rnd in range(3):
        for composed in (True, False):
            sigkernel_amd.routes.no_merged_loss = composed
            Xg = X.clone().requires_grad_(True)
            for _ in range(4): Xg.grad = None; step(Xg)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n_e): Xg.grad = None; vals[composed] = step(Xg)
            torch.cuda.synchronize()
            res.setdefault(("eager", composed), []).append((time.perf_counter() - t0) / n_e * 1e3)
            grads[composed] = Xg.grad.clone()
            sX = X.clone().requires_grad_(True)
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3): step(sX); sX.grad = None
            torch.cuda.current_stream().wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr): step(sX)
            for _ in range(3): gr.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n_g): gr.replay()
            torch.cuda.synchronize()
            res.setdefault(("graph", composed), []).append((time.perf_counter() - t0) / n_g * 1e3)
            assert torch.equal(sX.grad, grads[composed])
    gerr = float((grads[True] - grads[False]).abs().max() / grads[True].abs().max())
    verr = abs(float(vals[True]) - float(vals[False]))
    print("%-6s A=B=%3d len %3d dim %2d d=%d | eager: composed %.3f ms, merged %.3f | hipGraph: composed %.3f, merged %.3f | value diff %.1e, gradient rel diff %.1e"
          % (kern, A, M, D, d, min(res[("eager", True)]), min(res[("eager", False)]), min(res[("graph", True)]), min(res[("graph", False)]), verr, gerr), flush=True)
