import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import sigkernel_amd
from oracle import oracle as O
def walk(g, A, M, D, dt): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt)
d, A, B, M, N, D = 2, 3, 4, 150, 200, 12
g = torch.Generator().manual_seed(1)
X, Y = walk(g, A, M, D, torch.float64), walk(g, B, N, D, torch.float64)
w = torch.randn(A, B, generator=g, dtype=torch.float64)
k = sigkernel_amd.LinearKernel()
want = O.gram_grad_weighted(X, Y, w.numpy(), k, d, nthreads=8)
sk = sigkernel_amd.SigKernel(k, d)
for i in range(5):
    Xg = X.cuda().requires_grad_(True)
    (sk.compute_Gram(Xg, Y.cuda()) * w.cuda()).sum().backward()
    got = Xg.grad.cpu().numpy()
    print("rel err vs oracle %.3e" % (np.abs(got - want).max() / np.abs(want).max()), "worst rows", np.argsort(np.abs(got - want).max(axis=(0, 2)))[-4:])
