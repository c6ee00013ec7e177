import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigkernel_amd import _lib
P, Mc, Nc, d = 131072, 127, 127, 1
be = _lib.HipBackend(); ld = _lib._padded_ld(Nc, 8)
buf = torch.zeros(P, Mc, ld, device="cuda", dtype=torch.float64); buf[..., :Nc] = torch.randn(P, Mc, Nc, device="cuda", dtype=torch.float64) * 0.01
inc = buf[..., :Nc]
def t(f):
    for _ in range(2): f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]; ev[0].record()
    for i in range(3): f(); ev[i+1].record()
    torch.cuda.synchronize(); return min(ev[i].elapsed_time(ev[i+1]) for i in range(3))
print("fwd plain  %.3f ms" % t(lambda: be.solve_fwd(inc, d, flags=_lib.FLAG_FAST_ONLY)))
print("fwd edges  %.3f ms" % t(lambda: be.solve_fwd(inc, d, flags=_lib.FLAG_FAST_ONLY, want_edges=True)))
for w in (2, 4):
    os.environ["SK_WAVE_WPC"] = str(w)
    print("WPC=%d fwd plain  %.3f ms" % (w, t(lambda: be.solve_fwd(inc, d, flags=_lib.FLAG_FAST_ONLY))))
    print("WPC=%d fwd edges  %.3f ms" % (w, t(lambda: be.solve_fwd(inc, d, flags=_lib.FLAG_FAST_ONLY, want_edges=True))))
