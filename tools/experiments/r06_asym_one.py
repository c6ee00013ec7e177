import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
A=128; kind=sys.argv[1]; D=int(sys.argv[2]); d=int(sys.argv[3]); M=int(sys.argv[4]); N=int(sys.argv[5])
w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
sk = sigkernel_amd.SigKernel(k, d)
X, Y = walk(A, M, D), walk(A, N, D)
for _ in range(12):
    Xg = X.clone().requires_grad_(True)
    (sk.compute_Gram(Xg, Y) * w).sum().backward()
torch.cuda.synchronize()
