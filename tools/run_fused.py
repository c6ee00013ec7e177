#!/usr/bin/env python3
"""Run the fused-linear forward kernel on the headline Gram a few times (for rocprofv3 passes)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigkernel_amd import _lib
A = B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M, D, d = 128, 8, 1
g = torch.Generator().manual_seed(0)
mk = lambda n: (torch.cumsum(torch.randn(n, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = mk(A), mk(B)
be = _lib.HipBackend()
for _ in range(4):
    K = be.solve_fwd_fused_linear(X, Y, 1.0, d, False, True)
torch.cuda.synchronize()
print("ok", float(K[0, 0]))
