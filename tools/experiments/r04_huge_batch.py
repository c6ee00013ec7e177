#!/usr/bin/env python3
"""More than 2^31 pairs in one compute_Gram call (46400 x 46400 paths of 16 points): the fused launchers' 32-bit pair indices refuse
it, the host layer must tile -- values against the oracle on a sample, no overflow anywhere."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from oracle import oracle as O
A = int(sys.argv[1]) if len(sys.argv) > 1 else 46400
g = torch.Generator().manual_seed(3)
def walk(A, M, D): return torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) * (0.6 / np.sqrt(M * D))
for kern, name in ((sigkernel_amd.LinearKernel(), "linear"), (sigkernel_amd.RBFKernel(1.0), "rbf")):
    X, Y = walk(A, 16, 2), walk(A, 16, 2)
    sk = sigkernel_amd.SigKernel(kern, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = sk.compute_Gram(X.cuda(), Y.cuda())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert K.shape == (A, A)
    rows = [0, 1, A // 2, 46340 if A > 46340 else A - 2, A - 1]
    cols = [0, 7, A // 3, A - 1]
    want = O.gram_forward(X[rows], Y[cols], kern, 0)
    got = K[rows][:, cols].cpu().numpy()
    err = float(np.abs(got - want).max() / np.abs(want).max())
    fin = bool(torch.isfinite(K[-64:]).all()) and bool(torch.isfinite(K[:64]).all())
    print("%s: %d x %d = %.3e pairs in %.2f s, sample rel err %.1e, finite edges %s, max mem %.1f GB" % (name, A, A, float(A) * A, dt, err, fin, torch.cuda.max_memory_allocated() / 1e9), flush=True)
    assert err < 1e-11 and fin
    del K
