import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import sigkernel_amd
from sigkernel_amd import sigkernel as S
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def t(f, n=2, reps=3):
    for _ in range(1): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2], r
for D, A in ((20, 256), (20, 384), (20, 512), (12, 512)):
    X = walk(A, 64, D)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
    def step():
        Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Xg, sym=True).sum().backward(); return Xg.grad
    S._SYM_MIN_CELLS = None; ta, ga = t(step)
    S._SYM_MIN_CELLS = 0.0; tb, gb = t(step)
    S._SYM_MIN_CELLS = None
    print("rbf dim %d, %d paths of 64 points, sym Gram + backward: default %.3f ms, blocked triangle %.3f ms, rel diff %.2e" % (D, A, ta, tb, float((ga - gb).abs().max() / ga.abs().max())), flush=True)
