#!/usr/bin/env python3
"""Every kernel family with a second stream kept busy by small launches (what another tenant of the chip looks like): results bit
for bit, and the slowdown.  The fused forwards draw pairs from a work queue; the streaming solver / adjoint, the derivative solver
and the one-band fused adjoints give the waves of a SIMD shares by age rank (blockIdx / #CU) -- this measures what that costs when
the dispatch order is disturbed."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
g = torch.Generator().manual_seed(5)
def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) * (0.6 / np.sqrt(M * D))).cuda()
lin, rbf = sigkernel_amd.LinearKernel(), sigkernel_amd.RBFKernel(1.0)

def gram(sk, X, Y):
    return lambda: sk.compute_Gram(X, Y)
def gram_bwd(sk, X, Y, w):
    def f():
        Xg = X.clone().requires_grad_(True)
        (sk.compute_Gram(Xg, Y) * w).sum().backward()
        return Xg.grad
    return f
def kgrad(sk, X, Y, gam):
    return lambda: torch.stack(sk.compute_kernel_and_derivatives_Gram(X, Y, gam))

CASES = []
X, Y = walk(512, 128, 8), walk(512, 128, 8)
w = torch.randn(512, 512, generator=g, dtype=torch.float64).cuda()
CASES.append(("F1 fused forward (work queue), C3", gram(sigkernel_amd.SigKernel(lin, 1), X, Y)))
CASES.append(("A1 fused linear adjoint (chunks by age rank), C3 + backward", gram_bwd(sigkernel_amd.SigKernel(lin, 1), X, Y, w)))
X4, Y4 = walk(512, 64, 4), walk(512, 64, 4)
CASES.append(("A1 fused rbf adjoint (chunks by age rank), 512 x 512 of C4's shape + backward", gram_bwd(sigkernel_amd.SigKernel(rbf, 2), X4, Y4, w)))
X20, Y20 = walk(256, 128, 20), walk(256, 128, 20)
w2 = torch.randn(256, 256, generator=g, dtype=torch.float64).cuda()
CASES.append(("S1 streaming forward (shares by age rank), dim 20", gram(sigkernel_amd.SigKernel(lin, 1), X20, Y20)))
CASES.append(("S1 + S2 streaming forward + adjoint, dim 20 + backward", gram_bwd(sigkernel_amd.SigKernel(lin, 1), X20, Y20, w2)))
Xd, Yd, gam = walk(256, 128, 4), walk(256, 128, 4), walk(256, 128, 4)
CASES.append(("D2 fused derivative solver, 256 x 256, len 128", kgrad(sigkernel_amd.SigKernel(lin, 1), Xd, Yd, gam)))
Xm, Ym = walk(64, 600, 6), walk(64, 600, 6)
CASES.append(("F2 multi-band forward (queue), 64 x 64, len 600", gram(sigkernel_amd.SigKernel(rbf, 1), Xm, Ym)))

side = torch.cuda.Stream()
buf = torch.zeros(1 << 16, dtype=torch.float32, device="cuda")      # 256 KB: each launch occupies a few CUs for microseconds
# 2000 small launches as ONE graph: a replay is enqueued in microseconds and keeps the side stream busy for milliseconds, so the
# stream can be loaded for the whole timed region before it starts (a Python loop of launches drains as fast as it is enqueued)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): buf.add_(1.0)
    noise = torch.cuda.CUDAGraph()
    heavy = len(sys.argv) > 1 and sys.argv[1] == "heavy"      # a tenant that wants the whole chip: 4096^3 fp32 matrix products
    if heavy:
        Ma, Mb = torch.randn(4096, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
        for _ in range(3): Mc = Ma @ Mb
    with torch.cuda.graph(noise, stream=side):
        if heavy:
            for _ in range(20): Mc = Ma @ Mb
        else:
            for _ in range(2000): buf.add_(1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter(); noise.replay(); side.synchronize(); noise_s = time.perf_counter() - t0
print("side-stream load: graph of %s, %.2f ms per replay" % ("20 fp32 matrix products 4096^3" if heavy else "2000 small launches", noise_s * 1e3), flush=True)
def timed(f, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.current_stream().synchronize()
    return (time.perf_counter() - t0) / reps, r
def timed_busy(f, reps, expect_s):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(int(2.5 * expect_s / noise_s) + 2): noise.replay()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.current_stream().synchronize()
    dt = (time.perf_counter() - t0) / reps
    still = not side.query()
    side.synchronize()
    return dt, r, still
for name, f in CASES:
    for _ in range(5): r0 = f()
    one, _ = timed(f, 3)
    reps = max(3, min(20, int(0.15 / one)))
    alone, r0 = min((timed(f, reps) for _ in range(3)), key=lambda r: r[0])
    res = [timed_busy(f, reps, alone * reps) for _ in range(3)]
    busy, r1, still = min(res, key=lambda r: r[0])
    print("%-90s alone %8.3f ms  busy %8.3f ms  %+5.1f %%  bit-identical %s  (side stream busy throughout: %s)"
          % (name, alone * 1e3, busy * 1e3, (busy / alone - 1) * 100, torch.equal(r0, r1), all(r[2] for r in res)), flush=True)
