#!/usr/bin/env python3
"""GPU box: time the C3 forward (512 x 512 pairs, len 128, dim 8, d = 1, LinearKernel) through an alternative build of the
library (experiments on the fused kernel; results are NOT checked -- some variants compute nonsense on purpose).
usage: r02_variant_bench.py lib.so [lib.so ...]"""
import os, sys, subprocess
if len(sys.argv) > 2:
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, lib])
    sys.exit(0)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
gen = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)).cuda()
# SK_SHAPE="A,B,len,dim,dyadic,kind" (kind: lin / rbf); default the C3 headline
A, B, M, D, dy, kind = (os.environ.get("SK_SHAPE") or "512,512,128,8,1,lin").split(",")
X, Y = walk(int(A), int(M), int(D)), walk(int(B), int(M), int(D))
sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel() if kind == "lin" else sigkernel_amd.RBFKernel(1.0), int(dy))
for _ in range(3): K = sk.compute_Gram(X, Y)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K = sk.compute_Gram(X, Y); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
if os.environ.get("SK_EXP_DUMP"):
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(_lib.LIB_PATH).sk_exp_dump(int(os.environ.get("SK_EXP_DUMP")))
print("%-40s median %.3f ms  min %.3f ms   K[0,0] = %r  sum = %r" % (os.path.basename(sys.argv[1]), float(np.median(ts)), min(ts), float(K[0, 0]), float(K.sum())), flush=True)
