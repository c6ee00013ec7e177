#!/usr/bin/env python3
"""compute_Gram(X, Y) with a gradient at BASELINE configs[4]'s shape (256 x 256 pairs, len 512, dim 16, fp32, dyadic 2), 2 steps: for rocprofv3."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
g = torch.Generator().manual_seed(0)
mk = lambda: (torch.cumsum(torch.randn(256, 512, 16, generator=g, dtype=torch.float64), 1) / np.sqrt(512 * 16)).float().cuda()
X, Y = mk(), mk()
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2)
w = torch.randn(256, 256, generator=g).cuda()
for _ in range(2):
    Xg = X.clone().requires_grad_(True)
    (sk.compute_Gram(Xg, Y) * w).sum().backward()
torch.cuda.synchronize()
