import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigkernel_amd import _lib
A = B = 512; M, D, d = 128, 8, 1
g = torch.Generator().manual_seed(0)
mk = lambda n: (torch.cumsum(torch.randn(n, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = mk(A), mk(B); be = _lib.HipBackend()
def t(f):
    for _ in range(2): f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]; ev[0].record()
    for i in range(5): f(); ev[i+1].record()
    torch.cuda.synchronize(); return min(ev[i].elapsed_time(ev[i+1]) for i in range(5))
for w in (8, 9, 10, 8, 10, 9, 12, 10):
    os.environ["SK_FUSED_WPC"] = str(w)
    print("fused WPC=%d : %.3f ms" % (w, t(lambda: be.solve_fwd_fused_linear(X, Y, 1.0, d, False, True))))
