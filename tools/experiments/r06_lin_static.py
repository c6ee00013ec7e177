import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
be = _lib.get_backend()
def t(f, n=5, reps=5):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2], r
for D in (2, 4, 8, 12, 20, 32):
    X, Y = walk(256, 64, D), walk(256, 64, D)
    ms, inc = t(lambda: be.static_increments(0, 1.0, X, Y, True))
    print("linear static increments dim %d: %.3f ms checksum %.12g" % (D, ms, float(inc.sum())), flush=True)
