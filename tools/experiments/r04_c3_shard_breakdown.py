#!/usr/bin/env python3
"""Where a 64-row shard of the headline Gram (C3 strong scaling at 8 ranks) spends its 0.67 ms: the fused kernel alone, the
single-GPU API call, and the sharded call (collectives replaced by local copies, tools/shard_times.py's shim)."""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib, distributed as D
from sigkernel_amd.sigkernel import _fused_forward
g = torch.Generator().manual_seed(0)
mk = lambda A: (torch.cumsum(torch.randn(A, 128, 8, generator=g, dtype=torch.float64), 1) / 32).cuda()
X, Y = mk(512), mk(512)
be = _lib.get_backend()
sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1)
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rows in (512, 256, 128, 64):
    Xr = X[:rows].contiguous()
    k = timeit(lambda: _fused_forward(be, sk.static_kernel, Xr, Y, 1, False, gram=True))
    a = timeit(lambda: sk.compute_Gram(Xr, Y))
    print("rows %3d: staging + fused kernel %.3f ms | compute_Gram %.3f ms | ideal from 512 rows %.3f" % (rows, k, a, 0), flush=True)
state = {"rank": 0, "world": 8}
shim = types.SimpleNamespace(get_world_size=lambda group=None: state["world"], get_rank=lambda group=None: state["rank"], is_initialized=lambda: True,
                             get_backend=lambda group=None: "shim", ReduceOp=D.dist.ReduceOp, group=D.dist.group)
def fake_gather(out, inp, group):
    n = inp.shape[0]
    for r in range(state["world"]): out[r * n:(r + 1) * n].copy_(inp)
D.dist = shim; D._gather = fake_gather
skd = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1, process_group="shim")
print("sharded call, rank 0 of 8 (8 local copies standing in for the all-gather): %.3f ms" % timeit(lambda: skd.compute_Gram(X, Y)))
D._gather = lambda out, inp, group: None
print("sharded call without any gather traffic: %.3f ms" % timeit(lambda: skd.compute_Gram(X, Y)))
