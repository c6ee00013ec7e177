#!/usr/bin/env python3
"""compute_Gram(X, X, sym=True).sum().backward() at dyadic 1, dim <= 4: the triangle with the second-argument sums (k_adj_fused_rbf<1,2,*,4,YSIDE>,
which spills 168-176 bytes) against all pairs through the plain fused adjoint (no spill, twice the pairs)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import sigkernel as S
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def ms(f, n=5):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
orig = S._sym_triangle_ok
for A, M, D, d in ((256, 64, 3, 1), (512, 64, 3, 1), (1024, 64, 4, 1), (512, 128, 2, 1), (512, 64, 4, 2), (1024, 64, 4, 2), (2048, 64, 4, 2), (1024, 40, 2, 2), (512, 64, 3, 0), (1024, 100, 4, 0)):
    X = walk(A, M, D)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), d)
    w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda(); w = w + w.t()
    def step():
        Xg = X.clone().requires_grad_(True); (sk.compute_Gram(Xg, Xg, sym=True) * w).sum().backward(); return Xg.grad
    S._sym_triangle_ok = orig
    t_tri, g1 = ms(step), step()
    S._sym_triangle_ok = lambda *a: False
    t_all, g2 = ms(step), step()
    S._sym_triangle_ok = orig
    print("rbf dim %d d=%d, %4d paths of %3d points: triangle + second-argument sums %8.3f ms | all pairs, plain adjoint %8.3f ms | ratio %.2f | grad diff %.1e"
          % (D, d, A, M, t_tri, t_all, t_tri / t_all, float((g1 - g2).abs().max() / g2.abs().max())), flush=True)
