#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import _lib
from sigkernel_amd.sigkernel import _increments
be = _lib.get_backend(); dev = "cuda:0"
def walk(gen, A, M, D, dtype=torch.float64):
    return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)).to(dtype)
def run(kind, d, D, A, B, M, N, gram=True, dt=torch.float64, seed=0):
    gen = torch.Generator().manual_seed(seed)
    X, Y = (walk(gen, A, M, D, dt) * 1.5).to(dev), (walk(gen, B, N, D, dt) * 1.5).to(dev)
    sk = sigkernel_amd.LinearKernel(0.9) if kind == 0 else sigkernel_amd.RBFKernel(0.8)
    par = (1.0 if gram else 0.9) if kind == 0 else 0.8
    K = be.solve_fwd_fused_static(kind, par, X, Y, d, False, gram)
    if K is None:
        return None
    want = be.solve_fwd(_increments(be, sk, X.double(), Y.double(), gram), d)
    e = ((K.double() - want).abs() / want.abs().max())
    return float(e.max()), e.reshape(-1).cpu().numpy()
for kind in (0, 1):
    for d in (0, 1, 2):
        for D in (3, 8, 12, 16):
            for M in (20, (64 * (4 >> d)) + 1, (64 * (4 >> d)) + 2, 150 * (4 >> d), 290 * (4 >> d) // 2):
                for N in (158, 169, 414):
                    r = run(kind, d, D, 2, 5, M, N)
                    if r is None:
                        print("kind %d d %d D %2d M %4d N %4d: unsupported" % (kind, d, D, M, N)); continue
                    flag = "" if r[0] <= 1e-11 else "  <<<<<< " + " ".join("%.1e" % v for v in r[1])
                    print("kind %d d %d D %2d M %4d N %4d: %.2e%s" % (kind, d, D, M, N, r[0], flag))
