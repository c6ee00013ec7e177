import os, sys, time, faulthandler
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import sigkernel_amd
from sigkernel_amd import sigkernel as S
S._MERGED_MAX_PAIRS = 1 << 40
g = torch.Generator().manual_seed(0)
def walk(A, M, D, dt=torch.float64): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()
A, M, D, d = 64, 128, 8, 1
X, Y = walk(A, M, D), walk(A, M, D)
sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), d)
def step(Xg):
    v = sk.compute_mmd(Xg, Y); v.backward(); return v.detach()
for composed in (True, False):
    sigkernel_amd.routes.no_merged_loss = composed
    print("eager composed=%s" % composed, flush=True)
    Xg = X.clone().requires_grad_(True)
    for i in range(6):
        Xg.grad = None; v = step(Xg); torch.cuda.synchronize(); print(i, float(v), float(Xg.grad.abs().max()), flush=True)
    print("graph composed=%s" % composed, flush=True)
    sX = X.clone().requires_grad_(True)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step(sX); sX.grad = None
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize(); print("warm", flush=True)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): step(sX)
    print("captured", flush=True)
    for i in range(5): gr.replay(); torch.cuda.synchronize(); print("replay", i, float(sX.grad.abs().max()), flush=True)
print("done")
