#!/usr/bin/env python3
"""Forward-with-edges and adjoint-from-edges timed separately (GPU box).
usage: python tools/time_split.py [pairs] [Mc] [Nc] [dyadic]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigkernel_amd import _lib
P, Mc, Nc, d = [int(x) for x in (sys.argv[1:5] + ["131072", "127", "127", "1"][len(sys.argv) - 1:])]
be = _lib.HipBackend(); ld = _lib._padded_ld(Nc, 8)
buf = torch.zeros(P, Mc, ld, device="cuda", dtype=torch.float64)
buf[..., :Nc] = torch.randn(P, Mc, Nc, device="cuda", dtype=torch.float64) * 0.01
inc = buf[..., :Nc]
def t(f):
    for _ in range(2): f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]; ev[0].record()
    for i in range(3): f(); ev[i + 1].record()
    torch.cuda.synchronize(); return min(ev[i].elapsed_time(ev[i + 1]) for i in range(3))
_, edges = be.solve_fwd_keep_edges(inc, d)
print("P=%d %dx%d d=%d  WAVE_WPB=%s ADJ_WPB=%s: fwd+edges %.3f ms   adjoint from edges %.3f ms" % (
    P, Mc, Nc, d, os.environ.get("SK_WAVE_WPB", "-"), os.environ.get("SK_ADJ_WPB", "-"),
    t(lambda: be.solve_fwd_keep_edges(inc, d)), t(lambda: be.solve_adj(inc, d, flags=_lib.FLAG_FAST_ONLY, edges=edges))))
