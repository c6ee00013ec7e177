#!/usr/bin/env python3
"""Time the forward solver kernel alone on a headline-shaped tile under different tuning knobs (GPU box).
usage: python tools/tune_fwd.py [pairs] [Mc] [Nc] [dyadic] [dtype]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigkernel_amd import _lib

P = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
Mc = int(sys.argv[2]) if len(sys.argv) > 2 else 127
Nc = int(sys.argv[3]) if len(sys.argv) > 3 else 127
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dt = torch.float32 if len(sys.argv) > 5 and sys.argv[5] == "f32" else torch.float64
be = _lib.HipBackend()
ld = _lib._padded_ld(Nc, 8 if dt == torch.float64 else 4)
buf = (torch.randn(P, Mc, ld, device="cuda", dtype=dt) * 0.01)
inc = buf[..., :Nc]
alg = P * (Mc * Nc + 1) * buf.element_size()
cells = P * (Mc << d) * (Nc << d)


def run(label, **env):
    for k, v in env.items():
        os.environ[k] = str(v)
    for _ in range(2):
        be.solve_fwd(inc, d, flags=_lib.FLAG_FAST_ONLY)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    for i in range(5):
        be.solve_fwd(inc, d, flags=_lib.FLAG_FAST_ONLY)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(5))
    print("%-28s %8.3f ms  %7.1f GB/s alg  %6.3f Tcell/s" % (label, ms, alg / ms / 1e6, cells / ms / 1e9))
    for k in env:
        os.environ.pop(k)


for wpc in (4, 8, 12):      # (the prefetch distance is fixed at 2 macro-steps since round 5: its other values were knob-only variants)
    run("WPC=%d" % wpc, SK_WAVE_WPC=wpc)
