#!/usr/bin/env python3
"""C4 compute_mmd + backward with different numbers of row blocks of the symmetric K_XX (sigkernel._SYM_TILES)."""
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
for T in [int(a) for a in sys.argv[1:]] or [8, 16, 32]:
    S._SYM_TILES = T
    for _ in range(3): gr = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): gr = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    if ref is None: ref = gr
    print("T=%d: %.1f ms/step, grad rel diff vs first %.2e" % (T, dt * 1e3, float((gr - ref).abs().max() / ref.abs().max())))
