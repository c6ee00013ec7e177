import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sigkernel_amd
gen = torch.Generator().manual_seed(1)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)).cuda()
for kern, dy, M, D in ((sigkernel_amd.LinearKernel(), 1, 64, 5), (sigkernel_amd.RBFKernel(0.8), 2, 33, 3), (sigkernel_amd.RBFKernel(0.8), 0, 100, 8)):
    X, Y = walk(200000, M, D), walk(200000, M, D)
    sk = sigkernel_amd.SigKernel(kern, dy)
    os.environ["SK_RANK_W"] = "34,33,33"; Ke = sk.compute_kernel(X, Y); torch.cuda.synchronize()
    os.environ.pop("SK_RANK_W"); K = sk.compute_kernel(X, Y); torch.cuda.synchronize()
    ts = []
    for w in ("34,33,33", None):
        if w: os.environ["SK_RANK_W"] = w
        else: os.environ.pop("SK_RANK_W", None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sk.compute_kernel(X, Y); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    Xg = X[:64].clone().requires_grad_(True)
    sk.compute_kernel(Xg, Y[:64]).sum().backward()
    print(type(kern).__name__, dy, M, D, "equal == ranked:", bool(torch.equal(K, Ke)), "ms equal %.2f ranked %.2f" % tuple(ts), "grad finite", bool(torch.isfinite(Xg.grad).all()))
