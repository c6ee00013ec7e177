#!/usr/bin/env python3
"""Re-measure the library's cost table on THIS box (GPU; under two minutes): for every crossover in sk_cost_query one shape at the
threshold, both alternatives timed back to back, alternating -- and which entries sit within noise (+-6 %: the box-to-box spread)
of flipping.  usage: python tools/crossovers.py  [> profiles/rNN_crossovers.txt]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import _lib, sigkernel as S
NOISE = 0.06
g = torch.Generator().manual_seed(0)
def walk(A, M, D, dt=torch.float64): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()
def ms(f, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def ab(fa, fb, n=6, rounds=3):
    fa(); fb(); fa(); fb()
    ta, tb = [], []
    for _ in range(rounds):
        ta.append(ms(fa, n)); tb.append(ms(fb, n))
    return min(ta), min(tb)
def report(name, what, a_label, ta, b_label, tb, chosen):
    r = ta / tb
    close = abs(r - 1.0) <= NOISE
    right = (ta <= tb) == (chosen == "a") or close
    print("%-28s = %-8g | %s\n    %-34s %8.3f ms | %-34s %8.3f ms | ratio %.2f | the table picks %s: %s%s"
          % (name, _lib.cost(name), what, a_label, ta, b_label, tb, r, a_label if chosen == "a" else b_label,
             "ok" if right else "WRONG SIDE on this box", "  (within noise of flipping)" if close else ""), flush=True)
R = sigkernel_amd.routes
def set_routes(**kw):
    for k, v in kw.items(): setattr(R, k, v)
    S._route_query.cache_clear()
print("# tools/crossovers.py on %s -- each line: the shape at the table's threshold, both sides timed alternately (best of three)" % torch.cuda.get_device_name(0))

# --- mb_min_eff: linear dim 12 (multi-band or streamed), d = 1, 256 x 256 pairs with a gradient; efficiency just above the threshold
def grad_step(sk, X, Y, w):
    def f():
        Xg = X.clone().requires_grad_(True); (sk.compute_Gram(Xg, Y) * w).sum().backward()
    return f
for name, kern, D, d, M, grad in (("mb_min_eff", "linear", 12, 1, 140, True), ("stream_one_strip_cells", "linear", 12, 1, 129, True),
                                  ("mb_min_eff_rbf16_forward", "rbf", 16, 2, 260, False)):
    A = 128
    X, Y, w = walk(A, M, D), walk(A, M, D), torch.randn(A, A, generator=g, dtype=torch.float64).cuda()
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel() if kern == "linear" else sigkernel_amd.RBFKernel(1.0), d)
    f = grad_step(sk, X, Y, w) if grad else (lambda: sk.compute_Gram(X, Y))
    def fa(): set_routes(no_stream=True, no_fused_mb=False); f()
    def fb(): set_routes(no_stream=False, no_fused_mb=True); f()
    ta, tb = ab(fa, fb)
    set_routes(no_stream=False, no_fused_mb=False)
    route = _lib.get_backend().route(_lib.OP_ADJOINT if grad else _lib.OP_FORWARD, 0 if kern == "linear" else 1, D, M, M, d, False, 8)
    report(name, "%s dim %d d=%d, %d x %d pairs of %d points%s" % (kern, D, d, A, A, M, ", with a gradient" if grad else ""),
           "multi-band fused", ta, "streamed", tb, "a" if route == _lib.ROUTE_FUSED_MB else "b")

# --- sym_min_cells / sym_tiles: a symmetric Gram on the streaming route, one block against the blocked triangle, at the threshold
D, M, d = 20, 48, 1
cells_pair = ((M - 1) << d) ** 2
A = int((_lib.cost("sym_min_cells") / cells_pair) ** 0.5) + 1
X = walk(A, M, D)
sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), d)
def fa(): S._SYM_MIN_CELLS = 0.0; sk.compute_Gram(X, X, sym=True)
def fb(): S._SYM_MIN_CELLS = 1e30; sk.compute_Gram(X, X, sym=True)
ta, tb = ab(fa, fb, n=3)
S._SYM_MIN_CELLS = None
report("sym_min_cells", "linear dim %d d=%d, compute_Gram(X, X, sym=True), %d paths of %d points (%.1e cells)" % (D, d, A, M, A * A * cells_pair),
       "blocked triangle (sym_tiles)", ta, "one block, all pairs", tb, "a")

# --- sym_min_cells as the loss wrappers' limit: merged block K(X, [X; Y]) against the composition, at the threshold
M, D, d = 64, 3, 1
A = int((_lib.cost("sym_min_cells") / (((M - 1) << d) ** 2)) ** 0.5)
X, Y = walk(A, M, D), walk(A, M, D)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), d)
def step():
    Xg = X.clone().requires_grad_(True); sk.compute_mmd(Xg, Y).backward()
def fa(): set_routes(no_merged_loss=False); S._SYM_MIN_CELLS = 1e30; step()
def fb(): set_routes(no_merged_loss=True); S._SYM_MIN_CELLS = None; step()
ta, tb = ab(fa, fb, n=3)
set_routes(no_merged_loss=False); S._SYM_MIN_CELLS = None
report("sym_min_cells", "rbf dim 3 d=1, compute_mmd + backward, %d + %d paths of 64 points (just below the limit)" % (A, A),
       "merged block K(X,[X;Y])", ta, "composition of three Grams", tb, "a")

# --- paired_merge_cells
M, D, d = 64, 4, 1
n = int(_lib.cost("paired_merge_cells") / (((M - 1) << d) ** 2)) - 1
X, Y = walk(n, M, D), walk(n, M, D)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), d)
def stepd():
    Xg = X.clone().requires_grad_(True); sk.compute_distance(Xg, Y).backward()
def fa(): S._PAIRED_MERGE_CELLS = 1e30; stepd()
def fb(): S._PAIRED_MERGE_CELLS = 0.0; stepd()
ta, tb = ab(fa, fb, n=3)
S._PAIRED_MERGE_CELLS = None
report("paired_merge_cells", "rbf dim 4 d=1, compute_distance + backward, %d pairs of 64 points (just below the limit)" % n,
       "one batch of 2n pairs", ta, "three paired batches", tb, "a")

# --- sym_stream_min_paths: the symmetric forward on the streaming route (rbf dim 20), one block against the blocked triangle at the threshold
A = int(_lib.cost("sym_stream_min_paths"))
X = walk(A, 64, 20)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
def fa(): S._SYM_STREAM_MIN_PATHS = 0; sk.compute_Gram(X, X, sym=True)
def fb(): S._SYM_STREAM_MIN_PATHS = 1e9; sk.compute_Gram(X, X, sym=True)
ta, tb = ab(fa, fb, n=10)
S._SYM_STREAM_MIN_PATHS = None
report("sym_stream_min_paths", "rbf dim 20 d=1, compute_Gram(X, X, sym=True), %d paths of 64 points (streaming route)" % A, "blocked triangle", ta, "one block, all pairs", tb, "a")

# --- keep_increments_fraction: a one-tile Gram block on the streaming route, forward + backward, with the forward's increments kept or formed again
for kname, k in (("rbf", sigkernel_amd.RBFKernel(1.0)), ("linear", sigkernel_amd.LinearKernel())):
    X, Y = walk(256, 64, 20), walk(256, 64, 20)
    sk = sigkernel_amd.SigKernel(k, 1)
    def stepg():
        Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
    def fa(): S._KEEP_INCREMENTS_FRACTION = None; stepg()
    def fb(): S._KEEP_INCREMENTS_FRACTION = 0.0; stepg()
    ta, tb = ab(fa, fb, n=6)
    S._KEEP_INCREMENTS_FRACTION = None
    report("keep_increments_fraction", "%s dim 20 d=1, compute_Gram + backward, 256 x 256 pairs of 64 points (2.1 GB of increments)" % kname,
           "increments kept", ta, "formed again in backward", tb, "a")

# --- age-rank shares of mid-size no-queue launches, and bands on several waves (library knobs)
lib = _lib.load()
def knob(name, v):
    if v is None: os.environ.pop(name, None)
    else: os.environ[name] = v
    lib.sk_reload_knobs()
X, Y = walk(64, 128, 8), walk(512, 128, 8)
sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1)
def fa(): knob("SK_FUSED_MID", "1"); sk.compute_Gram(X, Y)
def fb(): knob("SK_FUSED_MID", "0"); sk.compute_Gram(X, Y)
ta, tb = ab(fa, fb, n=20)
knob("SK_FUSED_MID", None)
report("fused_mid_min_pairs_per_rank", "linear dim 8 d=1, 64 x 512 pairs of 128 points (a 64-row shard of the headline Gram)", "shares by wave age rank", ta, "equal shares", tb, "a")
X, Y = walk(32, 700, 3), walk(32, 700, 3)
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
def fa(): knob("SK_FUSEDMB_SPLIT", "1"); sk.compute_Gram(X, Y)
def fb(): knob("SK_FUSEDMB_SPLIT", "0"); sk.compute_Gram(X, Y)
ta, tb = ab(fa, fb, n=6)
knob("SK_FUSEDMB_SPLIT", None)
report("mb_split_max_resident_share", "rbf dim 3 d=1, 32 x 32 pairs of 700 points (1024 pairs: a third of the resident waves)", "bands on several waves", ta, "one wave per pair", tb, "a")
