#!/usr/bin/env python3
"""Sweep resident waves per CU of the streaming forward solver (SK_WAVE_WPC) on a synthetic tile.
usage: python tools/sweep_wpc.py pairs Mc Nc dyadic [f64|f32] wpc,wpc,..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigkernel_amd import _lib
P, Mc, Nc, d = [int(x) for x in sys.argv[1:5]]
dt = torch.float32 if sys.argv[5] == "f32" else torch.float64
wpcs = [int(x) for x in sys.argv[6].split(",")]
be = _lib.HipBackend(); ld = _lib._padded_ld(Nc, 8 if dt == torch.float64 else 4)
buf = torch.randn(P, Mc, ld, device="cuda", dtype=dt) * 0.01
inc = buf[..., :Nc]
def t(f):
    for _ in range(2): f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]; ev[0].record()
    for i in range(5): f(); ev[i+1].record()
    torch.cuda.synchronize(); return min(ev[i].elapsed_time(ev[i+1]) for i in range(5))
for w in wpcs:
    if w: os.environ["SK_WAVE_WPC"] = str(w)
    else: os.environ.pop("SK_WAVE_WPC", None)
    print("P=%d %dx%d d=%d %s WPC=%s : %.3f ms" % (P, Mc, Nc, d, sys.argv[5], w or "default", t(lambda: be.solve_fwd(inc, d, flags=_lib.FLAG_FAST_ONLY))))
