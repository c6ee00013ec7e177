#!/usr/bin/env python3
"""C4 compute_mmd + backward with the symmetric K_XX's row blocks on 1..4 alternating side streams (sigkernel._SYM_STREAMS)
and different block counts (sigkernel._SYM_TILES; doubled inside for big batches)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import sigkernel as S
A, M, D, d = 2048, 64, 4, 2
g = torch.Generator().manual_seed(0)
mk = lambda: (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = mk(), mk()
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), d)
def step():
    Xg = X.clone().requires_grad_(True)
    sk.compute_mmd(Xg, Y).backward()
    return Xg.grad
ref = None
for T in (8, 16):
    for n in (1, 2, 3, 4):
        S._SYM_TILES, S._SYM_STREAMS = T, n
        for _ in range(3): gr = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): gr = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if ref is None: ref = gr
        print("T=%d streams=%d: %.1f ms/step, grad rel diff vs first %.2e" % (T, n, dt * 1e3, float((gr - ref).abs().max() / ref.abs().max())), flush=True)
