"""compute_Gram(X, X, sym=True).sum().backward() on the streaming route (wide paths): the row blocks' increments kept for backward or formed again."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import sigkernel as S
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def t(f, n=3, reps=5):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2], r
for kname, k in (("rbf", sigkernel_amd.RBFKernel(1.0)), ("linear", sigkernel_amd.LinearKernel())):
    for A in (256, 512):
        X = walk(A, 64, 20)
        sk = sigkernel_amd.SigKernel(k, 1)
        def step():
            Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Xg, sym=True).sum().backward(); return Xg.grad
        S._KEEP_INCREMENTS_FRACTION = None; ta, ga = t(step)
        S._KEEP_INCREMENTS_FRACTION = 0.0; tb, gb = t(step)
        S._KEEP_INCREMENTS_FRACTION = None
        print("%-6s dim 20, %d paths of 64 points, sym Gram + backward: increments kept %.3f ms, formed again %.3f ms, same bits %s" % (kname, A, ta, tb, bool(torch.equal(ga, gb))), flush=True)
