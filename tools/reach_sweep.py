#!/usr/bin/env python3
"""Every route of the public API once (for `rocprofv3 --kernel-trace --stats`): static kernel x path dim x dyadic order x stencil x
precision x lengths x operation (Gram, symmetric Gram, paired batch -- each with and without a gradient --, the loss wrappers, the
derivative Gram, few pairs of long paths).  The kernel_stats.csv of that run is what tools/variants.py --reached reads: an instance no
call here launches is unreachable through sk_route_query.  usage (36 s natively; rocprofv3 needs hours for the same ~10^5 dispatches -- the HIP runtime's own launch log serves):
    AMD_LOG_LEVEL=3 python tools/reach_sweep.py 2>&1 | grep -o "ShaderName : .*" | grep "sk::" | sort | uniq -c > names.txt
    python tools/variants.py --reached names.txt > profiles/rNN_variants.txt"""
import itertools, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
def walk(g, A, M, D, dt): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()
g = torch.Generator().manual_seed(0)
shapes = ((8, 8), (20, 33), (33, 20), (64, 64), (65, 65), (100, 90), (128, 128), (129, 40), (40, 129), (130, 129), (200, 40), (257, 161), (300, 520), (64, 512), (512, 64))
if len(sys.argv) > 1: shapes = shapes[int(sys.argv[1])::int(sys.argv[2])]      # a slice of the shapes: reach_sweep.py <first> <stride>
n = 0
for kname, D, d, naive, dt in itertools.product(("linear", "rbf"), (1, 3, 4, 5, 8, 9, 16, 20), (0, 1, 2, 3), (False, True), (torch.float64, torch.float32)):
    k = sigkernel_amd.RBFKernel(0.9) if kname == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, d, _naive_solver=naive)
    for M, N in shapes:
        if d == 3 and max(M, N) > 130: continue
        A, B = 3, 4
        X, Y = walk(g, A, M, D, dt), walk(g, B, N, D, dt)
        sk.compute_Gram(X, Y); sk.compute_Gram(X, X, sym=True); sk.compute_kernel(X, Y[:A])
        Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
        Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Xg, sym=True).sum().backward()
        Xg = X.clone().requires_grad_(True); sk.compute_kernel(Xg, Y[:A]).sum().backward()
        if M == N:
            Xg = X.clone().requires_grad_(True); sk.compute_mmd(Xg, Y).backward()
            sk.compute_mmd(X, Y); sk.compute_scoring_rule(X, Y[:1]); sk.compute_distance(X, Y[:A])
            Xg = X.clone().requires_grad_(True); sk.compute_expected_scoring_rule(Xg, Y).backward()
        if not naive and max(M, N) <= 130 and d <= 2:
            sk.compute_kernel_and_derivatives_Gram(X, Y, torch.randn(A, M, D, generator=g).to(dt).cuda())
        n += 1
        if n % 50 == 0: print(n, "combinations", flush=True)
# a user-defined static kernel (the generic route: Gram_matrix in torch -> sk_increments -> solver -> sk_increments_adjoint) and the
# bit-exact kernels (the stored-grid rescue of the streaming adjoint runs them on exploding pairs)
class _Poly:
    def Gram_matrix(self, X, Y): return (1.0 + torch.einsum("amd,bnd->abmn", X, Y)) ** 2
    def batch_kernel(self, X, Y): return (1.0 + torch.einsum("amd,and->amn", X, Y)) ** 2
for dt in (torch.float64, torch.float32):
    for d in (0, 1):
        sk = sigkernel_amd.SigKernel(_Poly(), d)
        X, Y = walk(g, 3, 20, 3, dt), walk(g, 4, 17, 3, dt)
        sk.compute_Gram(X, Y); Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
        Xg = X.clone().requires_grad_(True); sk.compute_kernel(Xg, Y[:3]).sum().backward()
        sk.compute_kernel_and_derivatives_Gram(X, Y, torch.randn(3, 20, 3, generator=g).to(dt).cuda())
from sigkernel_amd import _lib
be = _lib.get_backend()
for dt in (torch.float64, torch.float32):
    inc = (torch.randn(6, 9, 16, generator=g, dtype=torch.float64) * 0.1).to(dt).cuda()[..., :11]
    be.solve_fwd(inc, 1, flags=_lib.FLAG_EXACT); be.solve_adj(inc, 1, flags=_lib.FLAG_EXACT)
    be.solve_deriv(torch.stack([inc, inc, inc]), 1, flags=_lib.FLAG_EXACT)
# big batches (work queue, age-rank shares, triangular blocks), long paths (bands on several waves)
for kname, D, d, A, M in (("linear", 8, 1, 512, 128), ("rbf", 4, 2, 512, 64), ("rbf", 3, 1, 128, 64), ("rbf", 16, 2, 64, 512), ("linear", 4, 0, 8, 2048), ("rbf", 3, 1, 4, 1500)):
    k = sigkernel_amd.RBFKernel(1.0) if kname == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, d)
    dt = torch.float32 if D == 16 else torch.float64
    X, Y = walk(g, A, M, D, dt), walk(g, A, M, D, dt)
    sk.compute_Gram(X, Y); sk.compute_Gram(X, X, sym=True)
    if M <= 512:
        Xg = X.clone().requires_grad_(True); sk.compute_mmd(Xg, Y).backward()
torch.cuda.synchronize()
print("%d shape combinations swept" % n)
