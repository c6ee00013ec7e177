import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("SKROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
g = torch.Generator().manual_seed(0)
def walk(A, M, D, dt=torch.float64): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()
def t(f, n=4, reps=5):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2], r
for kind in (os.environ.get("KINDS") or "rbf,linear").split(","):
    for D, dt in ((12, torch.float64), (20, torch.float64), (32, torch.float64), (20, torch.float32), (9, torch.float64)):
        k = sigkernel_amd.RBFKernel(1.0) if kind == "rbf" else sigkernel_amd.LinearKernel()
        sk = sigkernel_amd.SigKernel(k, 1)
        X, Y = walk(256, 64, D, dt), walk(256, 64, D, dt)
        def fwd(): return sk.compute_Gram(X, Y)
        def step():
            Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward(); return Xg.grad
        tf, K = t(fwd); ts, G = t(step)
        print("%-6s dim %2d %s 256 x 256 pairs of 64 points d=1 | forward %7.3f ms | forward + backward %7.3f ms | checksums %.12g %.12g" % (kind, D, "fp32" if dt == torch.float32 else "fp64", tf, ts, float(K.double().sum()), float(G.double().abs().sum())), flush=True)
