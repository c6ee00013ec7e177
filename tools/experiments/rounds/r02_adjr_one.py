import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigkernel_amd import _lib
be = _lib.get_backend(); dev = "cuda:0"
def walk(gen, A, M, D):
    return torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)
for (A, B, M, N, D, d) in [tuple(int(v) for v in c.split(",")) for c in os.environ["CASES"].split(";")]:
    gen = torch.Generator().manual_seed(1)
    X, Y = (walk(gen, A, M, D) * 2).to(dev), (walk(gen, B, N, D) * 2).to(dev)
    K, edges = be.solve_fwd_fused_rbf(X, Y, 0.9, d, False, True, keep_edges=True)
    torch.cuda.synchronize(); print("fwd ok", A, B, M, N, D, d, edges.numel(), flush=True)
    got = be.rbf_adjoint_fused(X, Y, 0.9, d, edges, None, gram=True)
    torch.cuda.synchronize(); print("adj ok", None if got is None else float(got[1]), flush=True)
    if got is None: continue
    inc = be.static_increments(1, 0.9, X, Y, True)
    _, W = be.solve_adj(inc, d, False, edges=edges)
    want = be.static_adjoint(1, 0.9, X, Y, W, None, True)
    print("  rel err %.3e" % float((got[0] - want).abs().max() / want.abs().max()), flush=True)
    if float((got[0] - want).abs().max() / want.abs().max()) > 1e-9:
        e = ((got[0] - want).abs() / want.abs().max())
        print("  per-row max err (a=0):", " ".join("%.0e" % v for v in e[0].max(dim=1).values.cpu().numpy()))
