#!/usr/bin/env python3
"""Time forward+adjoint (sk_solve_adj, fused kernel) on a synthetic tile (GPU box).
usage: python tools/tune_adj.py [pairs] [Mc] [Nc] [dyadic]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigkernel_amd import _lib

P = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
Mc = int(sys.argv[2]) if len(sys.argv) > 2 else 127
Nc = int(sys.argv[3]) if len(sys.argv) > 3 else 127
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1
be = _lib.HipBackend()
ld = _lib._padded_ld(Nc, 8)
buf = torch.zeros(P, Mc, ld, device="cuda", dtype=torch.float64)
buf[..., :Nc] = torch.randn(P, Mc, Nc, device="cuda", dtype=torch.float64) * 0.01
inc = buf[..., :Nc]
alg = P * (3 * Mc * Nc + 1) * 8
cells = P * (Mc << d) * (Nc << d)


def run(label, **env):
    for k, v in env.items():
        os.environ[k] = str(v)
    for _ in range(2):
        be.solve_adj(inc, d, flags=_lib.FLAG_FAST_ONLY)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for i in range(3):
        be.solve_adj(inc, d, flags=_lib.FLAG_FAST_ONLY)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(3))
    print("%-16s fwd(edges)+adj %8.3f ms  %7.1f GB/s alg(3x)  %6.3f Tcell/s" % (label, ms, alg / ms / 1e6, cells / ms / 1e9))
    for k in env:
        os.environ.pop(k)


run("default")
for wpc in [int(x) for x in os.environ.get("TUNE_WPC", "4,6,8").split(",")]:
    run("ADJ_WPC=%d" % wpc, SK_ADJ_WPC=wpc)
