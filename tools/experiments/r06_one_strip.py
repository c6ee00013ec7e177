import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib, sigkernel as S
g = torch.Generator().manual_seed(0)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
def t(f, n=3, reps=3):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2]
be = _lib.get_backend()
R = {0: "S", 1: "F", 2: "MB", 3: "MBs", 4: "Fs"}
A = 128
w = torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
for kind, D, d, Ms in (("linear", 8, 2, (70, 90, 110, 129)), ("linear", 3, 2, (70, 100, 129)), ("linear", 12, 2, (70, 100, 129)), ("linear", 12, 1, (70, 100, 129)), ("linear", 12, 0, (70, 100, 129, 200, 257)),
                       ("rbf", 3, 2, (70, 100)), ("rbf", 8, 2, (70, 100, 129)), ("rbf", 12, 1, (70, 100, 129)), ("rbf", 12, 0, (100, 129, 200, 257))):
    for M in Ms:
        k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
        sk = sigkernel_amd.SigKernel(k, d)
        X, Y = walk(A, M, D), walk(A, M, D)
        def fwd(): sk.compute_Gram(X, Y)
        def step():
            Xg = X.clone().requires_grad_(True); (sk.compute_Gram(Xg, Y) * w).sum().backward()
        res = []
        for ns in (False, True):
            sigkernel_amd.routes.no_stream = ns; S._route_query.cache_clear()
            kk = 0 if kind == "linear" else 1
            res.append((t(fwd), t(step), R[be.route(_lib.OP_FORWARD, kk, D, M, M, d, False, 8, ns)], R[be.route(_lib.OP_ADJOINT, kk, D, M, M, d, False, 8, ns)]))
        sigkernel_amd.routes.no_stream = False; S._route_query.cache_clear()
        print("%-6s dim %2d d=%d %3d points | default fwd %7.3f (%3s) fwd+bwd %8.3f (%3s) | no_stream fwd %7.3f (%3s) fwd+bwd %8.3f (%3s) | ratios %.2f %.2f"
              % (kind, D, d, M, res[0][0], res[0][2], res[0][1], res[0][3], res[1][0], res[1][2], res[1][1], res[1][3], res[1][0]/res[0][0], res[1][1]/res[0][1]), flush=True)
