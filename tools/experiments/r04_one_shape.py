#!/usr/bin/env python3
"""One compute_Gram shape, a few calls (for rocprofv3 passes): one_shape.py kind A M N D d [reps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
kind, A, M, N, D, d = sys.argv[1], *map(int, sys.argv[2:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = walk(A, M, D), walk(A, N, D)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0) if kind == "rbf" else sigkernel_amd.LinearKernel(), d)
for _ in range(reps): K = sk.compute_Gram(X, Y)
torch.cuda.synchronize()
