#!/usr/bin/env python3
"""Time the directional-derivative solver (sk_solve_deriv_*) on a synthetic tile, and the whole k_kgrad call.
usage: python tools/run_deriv.py [pairs] [Mc] [Nc] [dyadic] [f64|f32] [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import _lib

P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
Mc = int(sys.argv[2]) if len(sys.argv) > 2 else 127
Nc = int(sys.argv[3]) if len(sys.argv) > 3 else 127
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dt = torch.float32 if len(sys.argv) > 5 and sys.argv[5] == "f32" else torch.float64
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
es = 8 if dt == torch.float64 else 4
be = _lib.HipBackend()
ld = _lib._padded_ld(Nc, es)
buf = torch.randn(3, P, Mc, ld, device="cuda", dtype=dt) * 0.01
inc3 = buf[..., :Nc]
for flags, name in ((_lib.FLAG_FAST_ONLY, "k_deriv_wave"), (_lib.FLAG_SIMPLE, "k_deriv_simple")):
    be.solve_deriv(inc3, d, flags=flags)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        out = be.solve_deriv(inc3, d, flags=flags)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    byts = 3.0 * P * Mc * Nc * es
    cells = float(P) * (Mc << d) * (Nc << d)
    print("%-15s P=%d %dx%d d=%d %s: median %.3f ms  min %.3f ms  -> %.0f GB/s algorithmic (%.3f of 8 TB/s), %.3e cells/s"
          % (name, P, Mc, Nc, d, "f64" if es == 8 else "f32", ms[len(ms) // 2], ms[0], byts / ms[len(ms) // 2] / 1e6,
             byts / ms[len(ms) // 2] / 1e6 / 8000, cells / ms[len(ms) // 2] * 1e3))
    if P * Mc * Nc > 4e8 and flags == _lib.FLAG_FAST_ONLY and os.environ.get("SK_SKIP_SIMPLE"):
        break
print("ok", [float(o[0]) for o in out])
