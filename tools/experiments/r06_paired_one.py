import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
A = 262144
X, Y = (walk(A, 64, 3), walk(A, 64, 3)) if len(sys.argv) > 1 else (walk(A, 64, 8), walk(A, 64, 8))
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0) if len(sys.argv) > 1 else sigkernel_amd.LinearKernel(), 1)
for _ in range(6):
    Xg = X.clone().requires_grad_(True); sk.compute_kernel(Xg, Y).sum().backward()
torch.cuda.synchronize()
print("ppg", _lib.get_backend().last_fused_ppg)
