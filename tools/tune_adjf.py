#!/usr/bin/env python3
"""Time the fused linear adjoint (sk_linear_adjoint_fused_f64) against the unfused backward route on one Gram (GPU box).
usage: python tools/tune_adjf.py [A] [len] [dim] [dyadic]      (SK_ADJF_WPC / SK_ADJF_WPB: residency knobs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sigkernel_amd import _lib
A = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = int(sys.argv[2]) if len(sys.argv) > 2 else 128
D = int(sys.argv[3]) if len(sys.argv) > 3 else 8
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1
be = _lib.HipBackend()
g = torch.Generator().manual_seed(0)
walk = lambda n: (torch.cumsum(torch.randn(n, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = walk(A), walk(A)
go = torch.randn(A * A, generator=g, dtype=torch.float64).cuda()
def tm(f):
    for _ in range(2): f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]; ev[0].record()
    for i in range(3): f(); ev[i + 1].record()
    torch.cuda.synchronize(); return min(ev[i].elapsed_time(ev[i + 1]) for i in range(3))
K, edges = be.solve_fwd_fused_linear(X, Y, 1.0, d, False, True, keep_edges=True)
def old():
    inc = be.static_increments(0, 1.0, X, Y, gram=True)
    _, W = be.solve_adj(inc, d, False, edges=edges)
    return be.static_adjoint(0, 1.0, X, Y, W, go, True)
print("%dx%d len %d dim %d d=%d  WPC=%s WPB=%s: fused adjoint %.3f ms   (unfused backward %.3f ms)" % (
    A, A, M, D, d, os.environ.get("SK_ADJF_WPC", "-"), os.environ.get("SK_ADJF_WPB", "-"),
    tm(lambda: be.linear_adjoint_fused(X, Y, 1.0, d, edges, go)), tm(old)))
