#!/usr/bin/env python3
"""GPU box: unusual sizes through the round-2 kernels (multi-band fused forward, triangular symmetric launch, fused RBF adjoint)
against the streaming / unfused routes."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import _lib
be = _lib.get_backend(); dev = "cuda:0"
def walk(gen, A, M, D, dtype=torch.float64):
    return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)).to(dtype)
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
gen = torch.Generator().manual_seed(0)
ok = True
def check(name, err, tol):
    global ok
    flag = "ok" if err <= tol else "FAIL"
    if err > tol: ok = False
    print("%-70s %.2e %s" % (name, err, flag), flush=True)
# 1. very long second path, few pairs (multi-band, many units)
X, Y = walk(gen, 2, 700, 3).to(dev), walk(gen, 3, 5000, 3).to(dev)
for kern in (sigkernel_amd.RBFKernel(1.0), sigkernel_amd.LinearKernel()):
    sk = sigkernel_amd.SigKernel(kern, 1)
    K = sk.compute_Gram(X, Y); os.environ["SK_NO_FUSED_MB"] = "1"; sigkernel_amd.routes.reload(); K2 = sk.compute_Gram(X, Y); os.environ.pop("SK_NO_FUSED_MB"); sigkernel_amd.routes.reload()
    check("MB len 700 x 5000 d=1 %s" % type(kern).__name__, rel(K, K2), 1e-10)
# 2. paired, many pairs, long paths
X, Y = walk(gen, 3000, 300, 5).to(dev), walk(gen, 3000, 260, 5).to(dev)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.7), 2)
K = sk.compute_kernel(X, Y); os.environ["SK_NO_FUSED_MB"] = "1"; sigkernel_amd.routes.reload(); K2 = sk.compute_kernel(X, Y); os.environ.pop("SK_NO_FUSED_MB"); sigkernel_amd.routes.reload()
check("MB paired 3000 pairs len 300/260 d=2", rel(K, K2), 1e-10)
# 3. triangular launch, large batch (pair index beyond 2^22) and tiny ones
for A in (1, 2, 3, 3001):
    X = walk(gen, A, 24, 3).to(dev)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
    K = sk.compute_Gram(X, X, sym=True); K2 = sk.compute_Gram(X, X, sym=False)
    check("sym triangle A=%d" % A, rel(K, K2), 1e-12)
    assert torch.equal(K, K.t())
# 4. fused RBF adjoint: B = 1, A = 1, large A (several rounds of workgroups), paired
for (A, B, M, N, D, d) in ((1, 1, 64, 64, 4, 2), (5000, 1, 30, 40, 2, 1), (1, 3000, 50, 33, 3, 2), (4000, 3, 9, 120, 4, 1)):
    X, Y = walk(gen, A, M, D).to(dev), walk(gen, B, N, D).to(dev)
    w = torch.randn(A, B, generator=gen, dtype=torch.float64).to(dev)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.9), d)
    g = []
    for env in ("", "1"):
        if env: os.environ["SK_NO_FUSED_ADJOINT"] = "1"; sigkernel_amd.routes.reload()
        Xg = X.clone().requires_grad_(True)
        (sk.compute_Gram(Xg, Y) * w).sum().backward()
        g.append(Xg.grad)
        os.environ.pop("SK_NO_FUSED_ADJOINT", None); sigkernel_amd.routes.reload()
    check("fused RBF adjoint A=%d B=%d len %d/%d dim %d d=%d" % (A, B, M, N, D, d), rel(g[0], g[1]), 1e-9)
print("ALL OK" if ok else "FAILURES")
