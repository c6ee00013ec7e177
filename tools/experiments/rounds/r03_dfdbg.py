#!/usr/bin/env python3
"""Determinism / bit-identity probe of the fused derivative solver: per shape, distinct results over repeated runs of the fused route
(shifted and unshifted band layout) and of the unfused route, and the differences between them."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import _lib
def walk(g, A, M, D): return torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)
def run(kern, d, X, Y, gm, mode):
    os.environ.pop("SK_NO_FUSED_DERIV", None); os.environ.pop("SK_DERIVF_NOSHIFT", None)
    if mode == "unfused": os.environ["SK_NO_FUSED_DERIV"] = "1"
    if mode == "noshift": os.environ["SK_DERIVF_NOSHIFT"] = "1"
    sigkernel_amd.routes.reload()
    _lib.load().sk_reload_knobs()
    return sigkernel_amd.SigKernel(kern, d).compute_kernel_and_derivatives_Gram(X, Y, gm)
kname = sys.argv[1] if len(sys.argv) > 1 else "linear"
kern = sigkernel_amd.LinearKernel() if kname == "linear" else sigkernel_amd.RBFKernel(0.8)
for (d, A, B, M, N, D) in [(1,2,3,128,158,16),(1,2,3,100,158,16),(1,1,1,128,158,16),(1,1,1,60,300,16),(1,2,2,65,300,12),(1,2,3,128,158,8),(0,2,3,128,158,16),(2,2,3,70,170,9)]:
    g = torch.Generator().manual_seed(M + N)
    X, Y, gm = walk(g, A, M, D).cuda(), walk(g, B, N, D).cuda(), torch.randn(A, M, D, generator=g, dtype=torch.float64).cuda()
    res = {m: [run(kern, d, X, Y, gm, m) for _ in range(6)] for m in ("shift", "noshift", "unfused")}
    key = lambda r: tuple(torch.cat([t.flatten() for t in r]).tolist())
    nd = {m: len({key(r) for r in res[m]}) for m in res}
    ref = res["unfused"][0]
    diff = {m: max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(res[m][0], ref)) for m in ("shift", "noshift")}
    print((d, A, B, M, N, D), "distinct results", nd, "max rel diff vs unfused", diff, flush=True)
