#!/usr/bin/env python3
"""End-to-end timing of SigKernel.compute_kernel_and_derivatives_Gram.
usage: python tools/time_kgrad.py [A] [B] [len] [dim] [dyadic] [linear|rbf]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd

A = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
M = int(sys.argv[3]) if len(sys.argv) > 3 else 128
D = int(sys.argv[4]) if len(sys.argv) > 4 else 8
d = int(sys.argv[5]) if len(sys.argv) > 5 else 1
kern = sigkernel_amd.RBFKernel(1.0) if len(sys.argv) > 6 and sys.argv[6] == "rbf" else sigkernel_amd.LinearKernel()
gen = torch.Generator().manual_seed(0)
walk = lambda n: (torch.cumsum(torch.randn(n, M, D, generator=gen, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = walk(A), walk(B)
g = torch.randn(A, M, D, generator=gen, dtype=torch.float64).cuda()
sk = sigkernel_amd.SigKernel(kern, d)
for _ in range(2):
    out = sk.compute_kernel_and_derivatives_Gram(X, Y, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = sk.compute_kernel_and_derivatives_Gram(X, Y, g)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("k_kgrad %dx%d len %d dim %d d=%d %s: %.2f ms/call, %.3e entries/s" % (A, B, M, D, d, type(kern).__name__, dt * 1e3, A * B / dt))
