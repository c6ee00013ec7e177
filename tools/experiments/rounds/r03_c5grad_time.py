#!/usr/bin/env python3
"""compute_Gram(X, Y) with a gradient at BASELINE configs[4]'s shape (256 x 256 pairs, len 512, dim 16, fp32, dyadic 2): forward and
backward times with the multi-band fused adjoint and (SK_NO_FUSED_ADJOINT=1) on the unfused route, and the two gradients compared.
usage: r03_c5grad_time.py [batch] [len] [dim] [dtype]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
A = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
D = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dt = torch.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else torch.float32
g = torch.Generator().manual_seed(0)
mk = lambda: (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()
X, Y = mk(), mk()
w = torch.randn(A, A, generator=g).to(dt).cuda()
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2)
def step():
    Xg = X.clone().requires_grad_(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = sk.compute_Gram(Xg, Y); torch.cuda.synchronize(); t1 = time.perf_counter()
    (K * w).sum().backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return t1 - t0, t2 - t1, Xg.grad
out = {}
for tag, env in (("fused multi-band adjoint", None), ("unfused route", "SK_NO_FUSED_ADJOINT")):
    if env: os.environ[env] = "1"; sigkernel_amd.routes.reload()
    step()
    best = min((step() for _ in range(3)), key=lambda r: r[0] + r[1])
    out[tag] = best[2]
    print("%-26s fwd %.1f ms  bwd %.1f ms  total %.1f ms" % (tag, best[0] * 1e3, best[1] * 1e3, (best[0] + best[1]) * 1e3), flush=True)
    if env: del os.environ[env]; sigkernel_amd.routes.reload()
a, b = out["fused multi-band adjoint"].double(), out["unfused route"].double()
print("gradients: max rel diff %.2e" % float((a - b).abs().max() / b.abs().max()))
