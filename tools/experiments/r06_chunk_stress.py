#!/usr/bin/env python3
"""Random batch shapes through the fused adjoints (uneven chunks, several launches, the swapped second-argument route) against the streaming
route of the same call (routes.no_fused_adjoint): gradients to 1e-9, values to 1e-12.  GPU only, no oracle.  usage: r06_chunk_stress.py [n] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import sigkernel as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator().manual_seed(1)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
bad = 0
for it in range(n):
    kind = rng.choice(["linear", "rbf"]); d = int(rng.integers(0, 3)); D = int(rng.integers(1, 9))
    A = int(rng.choice([1, 2, 3, 5, 7, 31, 61, 97, 127, 129, 251, 300, 509, 640, 701])) if rng.random() < 0.7 else int(rng.integers(1, 700))
    B = int(rng.choice([1, 2, 3, 5, 7, 31, 61, 97, 127, 129, 251, 300, 509, 640, 701])) if rng.random() < 0.7 else int(rng.integers(1, 700))
    if A * B > 120000: B = max(1, 120000 // A)
    M = int(rng.integers(2, 66 if d == 2 else 130)); N = int(rng.integers(2, 66 if d == 2 else 130))
    if rng.random() < 0.25: M = int(rng.integers(130, 400))       # long first paths: the swapped route
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(float(rng.uniform(0.5, 2.0)))
    sk = sigkernel_amd.SigKernel(k, d)
    X, Y = walk(A, M, D), walk(B, N, D)
    w = torch.randn(A, B, generator=g, dtype=torch.float64).cuda()
    out = []
    for off in (False, True):
        sigkernel_amd.routes.no_fused_adjoint = off; sigkernel_amd.routes.no_adjoint_swap = off; S._route_query.cache_clear()
        Xg = X.clone().requires_grad_(True)
        K = sk.compute_Gram(Xg, Y); (K * w).sum().backward()
        out.append((K.detach(), Xg.grad))
    sigkernel_amd.routes.no_fused_adjoint = False; sigkernel_amd.routes.no_adjoint_swap = False; S._route_query.cache_clear()
    ek = float((out[0][0] - out[1][0]).abs().max() / out[1][0].abs().max())
    eg = float((out[0][1] - out[1][1]).abs().max() / max(float(out[1][1].abs().max()), 1e-300))
    if not (ek <= 1e-11 and eg <= 1e-8):
        bad += 1
        print("MISMATCH", kind, "D", D, "d", d, "A", A, "B", B, "M", M, "N", N, "value", ek, "grad", eg, flush=True)
# the loss wrappers: the one-launch route (rectangle K(X, [X; Y]) + triangle, ONE fused adjoint) against the composition of Gram calls
for it in range(n // 4):
    kind = rng.choice(["linear", "rbf"]); d = int(rng.integers(0, 3)); D = int(rng.integers(1, 9))
    A, B = int(rng.integers(2, 200)), int(rng.integers(2, 200)); M = int(rng.integers(2, 66 if d == 2 else 130))
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(float(rng.uniform(0.5, 2.0)))
    sk = sigkernel_amd.SigKernel(k, d)
    X, Y = walk(A, M, D), walk(B, M, D)
    out = []
    for off in (False, True):
        sigkernel_amd.routes.no_loss_launch = off; S._route_query.cache_clear()
        Xg = X.clone().requires_grad_(True)
        v = sk.compute_mmd(Xg, Y) if it % 2 else sk.compute_expected_scoring_rule(Xg, Y)
        v.backward()
        out.append((v.detach(), Xg.grad))
    sigkernel_amd.routes.no_loss_launch = False; S._route_query.cache_clear()
    ev = float((out[0][0] - out[1][0]).abs() / max(float(out[1][0].abs()), 1e-3))
    eg = float((out[0][1] - out[1][1]).abs().max() / max(float(out[1][1].abs().max()), 1e-300))
    if not (ev <= 1e-9 and eg <= 1e-8):
        bad += 1
        print("MISMATCH loss", kind, "D", D, "d", d, "A", A, "B", B, "M", M, "value", ev, "grad", eg, flush=True)
print("r06_chunk_stress: %d + %d cases, %d mismatches" % (n, n // 4, bad))
sys.exit(1 if bad else 0)
