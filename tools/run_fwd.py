#!/usr/bin/env python3
"""Run the forward solver kernel a few times on a synthetic tile (for rocprofv3 passes).
usage: python tools/run_fwd.py [pairs] [Mc] [Nc] [dyadic] [f64|f32] [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigkernel_amd import _lib

P = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
Mc = int(sys.argv[2]) if len(sys.argv) > 2 else 127
Nc = int(sys.argv[3]) if len(sys.argv) > 3 else 127
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dt = torch.float32 if len(sys.argv) > 5 and sys.argv[5] == "f32" else torch.float64
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
be = _lib.HipBackend()
ld = _lib._padded_ld(Nc, 8 if dt == torch.float64 else 4)
buf = torch.randn(P, Mc, ld, device="cuda", dtype=dt) * 0.01
inc = buf[..., :Nc]
for _ in range(reps):
    if os.environ.get('SK_RUN_EDGES'):
        out = be.solve_fwd(inc, d, flags=_lib.FLAG_FAST_ONLY, want_edges=True)[0]
    else:
        out = be.solve_fwd(inc, d, flags=_lib.FLAG_FAST_ONLY)
torch.cuda.synchronize()
print("ok", float(out[0]))
