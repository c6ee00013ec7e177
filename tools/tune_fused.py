#!/usr/bin/env python3
"""Sweep resident waves per CU of the fused forward kernels (SK_FUSED_WPC).
usage: python tools/tune_fused.py [A] [len] [dim] [dyadic] [wpc,wpc,...] [linear|rbf]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigkernel_amd import _lib
A = B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = int(sys.argv[2]) if len(sys.argv) > 2 else 128
D = int(sys.argv[3]) if len(sys.argv) > 3 else 8
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1
wpcs = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "0,8,10,12").split(",")]
kind = sys.argv[6] if len(sys.argv) > 6 else "linear"
g = torch.Generator().manual_seed(0)
mk = lambda n: (torch.cumsum(torch.randn(n, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = mk(A), mk(B); be = _lib.HipBackend()
def t(f):
    for _ in range(2): f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]; ev[0].record()
    for i in range(5): f(); ev[i+1].record()
    torch.cuda.synchronize(); return min(ev[i].elapsed_time(ev[i+1]) for i in range(5))
for w in wpcs:
    if w: os.environ["SK_FUSED_WPC"] = str(w)
    else: os.environ.pop("SK_FUSED_WPC", None)
    print(kind + " fused %dx%d len %d dim %d d=%d WPC=%s : %.3f ms" % (A, B, M, D, d, w or "default", t((lambda: be.solve_fwd_fused_rbf(X, Y, 1.0, d, False, True)) if kind == "rbf" else
                                                                  (lambda: be.solve_fwd_fused_linear(X, Y, 1.0, d, False, True)))))
