#!/usr/bin/env python3
"""A few training-sized steps for a rocprofv3 --kernel-trace pass: r05_small_steps.py <mmd A | gram A sym | shard rows> [reps]
   mmd A      compute_mmd(X, Y).backward(), A x A paths of BASELINE configs[1]'s shape (len 64, dim 3, rbf, d = 1)
   c2         compute_Gram(X, X, sym=True), 128 paths of that shape
   shard R    rows R of the headline config against all 512 (len 128, dim 8, linear, d = 1)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
what = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
if what == "mmd":
    X, Y = walk(n, 64, 3), walk(n, 64, 3)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
    def step():
        Xg = X.detach().requires_grad_(True)
        sk.compute_mmd(Xg, Y).backward()
elif what == "c2":
    X = walk(128, 64, 3)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
    def step(): sk.compute_Gram(X, X, sym=True)
else:
    X, Y = walk(n, 128, 8), walk(512, 128, 8)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1)
    def step(): sk.compute_Gram(X, Y)
for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): step()
torch.cuda.synchronize()
print("%s %d: %.1f us/step" % (what, n, (time.perf_counter() - t0) / reps * 1e6))
