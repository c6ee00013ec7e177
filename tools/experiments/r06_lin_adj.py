"""sk_static_adjoint (kind 0, 9..32 dims) alone: 256 x 256 pairs, random W."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
from sigkernel_amd import _lib
be = _lib.get_backend()
g = torch.Generator().manual_seed(0)
def t(f, n=4, reps=5):
    for _ in range(2): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[reps // 2], r
for (A, B, M, N, D, dt) in [(256, 256, 64, 64, 12, torch.float64), (256, 256, 64, 64, 20, torch.float64), (256, 256, 64, 64, 32, torch.float64),
                            (256, 256, 64, 64, 20, torch.float32), (128, 128, 128, 128, 20, torch.float64), (128, 128, 100, 100, 12, torch.float64)][slice(*([int(os.environ['ONLY']), int(os.environ['ONLY']) + 1] if os.environ.get('ONLY') else [None]))]:
    X = (torch.randn(A, M, D, generator=g, dtype=torch.float64) / D ** .5).to(dt).cuda(); Y = (torch.randn(B, N, D, generator=g, dtype=torch.float64) / D ** .5).to(dt).cuda()
    ld = -(-(N - 1) // 16) * 16
    W = torch.zeros(A, B, M - 1, ld, dtype=dt, device="cuda"); torch.manual_seed(1); W[..., :N - 1] = torch.randn(A, B, M - 1, N - 1, dtype=dt, device="cuda")
    Wv = W[..., :N - 1]
    sc = torch.randn(A, B, dtype=dt, device="cuda")
    ms, r = t(lambda: be.static_adjoint(int(os.environ.get('KIND', 0)), 1.0, X, Y, Wv, sc, True))
    gb = W.numel() * W.element_size() / 1e9
    print("%d x %d pairs, %d x %d points, dim %d %s: %.3f ms  (W %.2f GB -> %.2f TB/s)  checksum %.10g" % (A, B, M, N, D, "fp32" if dt == torch.float32 else "fp64", ms, gb, gb / ms, float(r.double().abs().sum())), flush=True)
