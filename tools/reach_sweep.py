#!/usr/bin/env python3
"""Every route of the public API once: static kernel x path dim x dyadic order x stencil x precision x lengths x operation (Gram,
symmetric Gram, paired batch -- each with and without a gradient --, the loss wrappers, the derivative Gram, few pairs of long paths),
then the calls whose kernel variants are gated by BATCH SIZE or by special lengths (work queue, age-rank shares, triangular blocks with
second-argument sums, bands on several waves, full multi-band efficiency, the derivative solver's unshifted bands).  The library counts
its own launches per kernel instance (sk_launch_trace, round 6: no profiler, no runtime log -- a minute natively): an instance no call
here launches is unreachable through sk_route_query / the exported entry points, and tests/test_abi.py fails on it.  usage:
    python tools/reach_sweep.py [first stride] [--out counts.txt]      (GPU box)   -> "count<TAB>device symbol" per instance launched
    python tools/variants.py --reached counts.txt > profiles/rNN_variants.txt     (CPU: the build's instances with their launches)
As a module: run(first=0, stride=1, check=0.0) -> {device symbol: launches}; check > 0: that share of the small Gram calls is compared
with the CPU oracle (values to 1e-9 / 1e-4 fp32, gradients to 1e-7 / 1e-3)."""
import itertools, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sigkernel_amd
from sigkernel_amd import _lib


def walk(g, A, M, D, dt):
    return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()


class _Poly:
    """a user-defined static kernel (the generic route: Gram_matrix in torch -> sk_increments -> solver -> sk_increments_adjoint)"""
    def Gram_matrix(self, X, Y): return (1.0 + torch.einsum("amd,bnd->abmn", X, Y)) ** 2
    def batch_kernel(self, X, Y): return (1.0 + torch.einsum("amd,and->amn", X, Y)) ** 2


SHAPES = ((8, 8), (20, 33), (33, 20), (64, 64), (65, 65), (100, 90), (128, 128), (129, 40), (40, 129), (130, 129), (200, 40), (257, 161), (300, 520),
          (64, 512), (512, 64))


def run(first=0, stride=1, check=0.0, verbose=True):
    _lib.launch_trace(True)
    _lib.launch_counts(reset=True)
    g = torch.Generator().manual_seed(0)
    rng = np.random.default_rng(0)
    O = None
    if check > 0:
        from oracle import oracle as O      # noqa: N811 -- the checker (test infrastructure), never the thing measured
    shapes = SHAPES[first::stride]
    n = checked = 0
    worst = 0.0
    for kname, D, d, naive, dt in itertools.product(("linear", "rbf"), (1, 3, 4, 5, 8, 9, 16, 20), (0, 1, 2, 3), (False, True), (torch.float64, torch.float32)):
        k = sigkernel_amd.RBFKernel(0.9) if kname == "rbf" else sigkernel_amd.LinearKernel()
        sk = sigkernel_amd.SigKernel(k, d, _naive_solver=naive)
        for M, N in shapes:
            if d == 3 and max(M, N) > 130: continue
            A, B = 3, 4
            X, Y = walk(g, A, M, D, dt), walk(g, B, N, D, dt)
            K = sk.compute_Gram(X, Y); sk.compute_Gram(X, X, sym=True); sk.compute_kernel(X, Y[:A])
            Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
            if O is not None and max(M, N) <= 130 and d <= 2 and rng.random() < check:
                tol = (1e-9, 1e-7) if dt == torch.float64 else (1e-4, 1e-3)
                want = O.gram_forward(X.cpu(), Y.cpu(), k, d, naive=naive)
                e = float(np.max(np.abs(K.double().cpu().numpy() - want)) / np.max(np.abs(want)))
                assert e <= tol[0], ("value", kname, D, d, naive, dt, M, N, e)
                gw = O.gram_grad_weighted(X.cpu(), Y.cpu(), np.ones((A, B)), k, d, naive=naive)
                eg = float(np.max(np.abs(Xg.grad.double().cpu().numpy() - gw)) / np.max(np.abs(gw)))
                assert eg <= tol[1], ("gradient", kname, D, d, naive, dt, M, N, eg)
                worst = max(worst, e / tol[0], eg / tol[1])
                checked += 1
            Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Xg, sym=True).sum().backward()
            Xg = X.clone().requires_grad_(True); sk.compute_kernel(Xg, Y[:A]).sum().backward()
            if M == N:
                Xg = X.clone().requires_grad_(True); sk.compute_mmd(Xg, Y).backward()
                sk.compute_mmd(X, Y); sk.compute_scoring_rule(X, Y[:1]); sk.compute_distance(X, Y[:A])
                Xg = X.clone().requires_grad_(True); sk.compute_expected_scoring_rule(Xg, Y).backward()
            if not naive and max(M, N) <= 130 and d <= 2:
                sk.compute_kernel_and_derivatives_Gram(X, Y, torch.randn(A, M, D, generator=g).to(dt).cuda())
            n += 1
            if verbose and n % 500 == 0: print(n, "combinations", flush=True)
    # a user-defined static kernel and the bit-exact kernels (the stored-grid rescue of the streaming adjoint runs them on exploding pairs)
    for dt in (torch.float64, torch.float32):
        for d in (0, 1):
            sk = sigkernel_amd.SigKernel(_Poly(), d)
            X, Y = walk(g, 3, 20, 3, dt), walk(g, 4, 17, 3, dt)
            sk.compute_Gram(X, Y); Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
            Xg = X.clone().requires_grad_(True); sk.compute_kernel(Xg, Y[:3]).sum().backward()
            sk.compute_kernel_and_derivatives_Gram(X, Y, torch.randn(3, 20, 3, generator=g).to(dt).cuda())
    be = _lib.get_backend()
    for dt in (torch.float64, torch.float32):
        inc = (torch.randn(6, 9, 16, generator=g, dtype=torch.float64) * 0.1).to(dt).cuda()[..., :11]
        be.solve_fwd(inc, 1, flags=_lib.FLAG_EXACT); be.solve_adj(inc, 1, flags=_lib.FLAG_EXACT)
        be.solve_deriv(torch.stack([inc, inc, inc]), 1, flags=_lib.FLAG_EXACT)
    if first == 0:
        gated(g, be)
    torch.cuda.synchronize()
    counts = _lib.launch_counts()
    if verbose:
        print("%d shape combinations swept, %d Gram matrices checked against the oracle (worst error / tolerance %.2g), %d kernel instances launched"
              % (n, checked, worst, len(counts)))
    return counts


def gated(g, be):
    """The variants a 3 x 4 batch cannot show."""
    f64, f32 = torch.float64, torch.float32
    RBF, LIN = sigkernel_amd.RBFKernel, sigkernel_amd.LinearKernel
    # big batches (work queue, age-rank shares, triangular blocks), long paths (bands on several waves)
    for kname, D, d, A, M in (("linear", 8, 1, 512, 128), ("rbf", 4, 2, 512, 64), ("rbf", 3, 1, 128, 64), ("rbf", 16, 2, 64, 512), ("linear", 4, 0, 8, 2048),
                              ("rbf", 3, 1, 4, 1500)):
        sk = sigkernel_amd.SigKernel(RBF(1.0) if kname == "rbf" else LIN(), d)
        dt = f32 if D == 16 else f64
        X, Y = walk(g, A, M, D, dt), walk(g, A, M, D, dt)
        sk.compute_Gram(X, Y); sk.compute_Gram(X, X, sym=True)
        if M <= 512:
            Xg = X.clone().requires_grad_(True); sk.compute_mmd(Xg, Y).backward()
    # the symmetric Gram WITH a gradient and enough grid cells for the blocked triangle (sym_min_cells, 5e9): second-argument sums of the
    # one-band RBF adjoint -- one coarse row per lane on fewer than 64 lanes at dyadic 1 / 2 (paths of <= 32 points), two on the full
    # wave at dyadic 0 (65..128 points), and the full-wave forms of BASELINE configs[3]'s shape ...
    for d, M, A in ((1, 32, 1280), (2, 32, 640), (0, 128, 640), (0, 40, 2048), (1, 64, 640), (2, 64, 320)):
        sk = sigkernel_amd.SigKernel(RBF(0.8), d)
        Xg = walk(g, A, M, 3, f64).requires_grad_(True)
        sk.compute_Gram(Xg, Xg, sym=True).sum().backward()
        del Xg
        torch.cuda.empty_cache()
    # ... and of the streaming route (sk_static_adjoint2: dyadic 3 is beyond every fused kernel), path dims 4 / 8 / 16 / 32 wide
    for kname, D, dt in itertools.product(("rbf", "linear"), (3, 8, 12, 20), (f64, f32)):
        if kname == "linear" and D > 8: continue      # (no second-argument kernel: such calls solve all pairs)
        sk = sigkernel_amd.SigKernel(RBF(0.8) if kname == "rbf" else LIN(), 3)
        Xg = walk(g, 1024, 10, D, dt).requires_grad_(True)
        sk.compute_Gram(Xg, Xg, sym=True).sum().backward()
        del Xg
        torch.cuda.empty_cache()
    # long first paths against short second ones with a gradient, LinearKernel: the one-band adjoint on (y, x) with second-argument sums,
    # on the full wave and on fewer lanes at each dyadic order (the 3 x 4 sweep above has no second paths of <= 33 points at dyadic 2)
    for d, (M, N) in itertools.product((0, 1, 2), ((200, 20), (300, 100), (400, 60))):
        sk = sigkernel_amd.SigKernel(LIN(), d)
        Xg = walk(g, 5, M, 6, f64).requires_grad_(True)
        sk.compute_Gram(Xg, walk(g, 7, N, 6, f64)).sum().backward()
    # ... and RBFKernel: the one-band adjoint's second-argument sums INSTEAD of the first-argument ones (dim 3: every dyadic order, two
    # rows per lane at dyadic 1; dim 6: dyadic 0 and 1)
    for D, d, (M, N) in itertools.product((3, 6), (0, 1, 2), ((200, 20), (300, 100), (400, 60))):
        if (D == 6 and (d == 2 or (d == 1 and N > 64))) or (d == 2 and N > 64): continue
        sk = sigkernel_amd.SigKernel(RBF(0.9), d)
        Xg = walk(g, 5, M, D, f64).requires_grad_(True)
        sk.compute_Gram(Xg, walk(g, 7, N, D, f64)).sum().backward()
    # paired batches of more pairs than resident lane groups, with a gradient: several pairs per lane group in the linear one-band
    # adjoint (PAIRED), on the full wave and on fewer lanes
    for kname, (d, M, Pn) in itertools.product(("linear", "rbf"), ((0, 128, 5000), (0, 40, 9000), (1, 128, 5000), (1, 40, 9000), (2, 64, 5000), (2, 30, 9000))):
        sk = sigkernel_amd.SigKernel(LIN() if kname == "linear" else RBF(0.9), d)
        Xg = walk(g, Pn, M, 4, f64).requires_grad_(True)
        sk.compute_kernel(Xg, walk(g, Pn, M, 4, f64)).sum().backward()
        del Xg
        torch.cuda.empty_cache()
    # paths of 25..32 dims with a gradient (the static adjoint's 32-dim instances; the sweep above has 20 dims: the 24-dim ones)
    for kname, dt, (M, N) in itertools.product(("rbf", "linear"), (f64, f32), ((40, 50), (30, 140))):
        sk = sigkernel_amd.SigKernel(RBF(0.9) if kname == "rbf" else LIN(), 1)
        Xg = walk(g, 3, M, 30, dt).requires_grad_(True)
        sk.compute_Gram(Xg, walk(g, 4, N, 30, dt)).sum().backward()
    # paths of 9..32 dims with a gradient: the tiled static adjoints (LinearKernel 16 / 24 / 32 dims, RBFKernel 24 / 32; second paths of
    # <= 64 and of 65..128 points)
    for kname, D, dt, N in itertools.product(("linear", "rbf"), (12, 20, 30), (f64, f32), (50, 100)):
        sk = sigkernel_amd.SigKernel(LIN() if kname == "linear" else RBF(0.9), 1)
        Xg = walk(g, 3, 40, D, dt).requires_grad_(True)
        sk.compute_Gram(Xg, walk(g, 4, N, D, dt)).sum().backward()
    # the fused derivative solver on first paths of 64 k + 1 points (bands that need no shifted lanes) against second paths of 126 points
    # and more; its in-LDS band boundary (two bands, 126..157-point second paths)
    for kname, d, (M, N) in itertools.product(("linear", "rbf"), (0, 1, 2), ((65, 130), (129, 140), (129, 200), (100, 140))):
        sk = sigkernel_amd.SigKernel(RBF(0.8) if kname == "rbf" else LIN(), d)
        X, Y = walk(g, 3, M, 3, f64), walk(g, 4, N, 3, f64)
        sk.compute_kernel_and_derivatives_Gram(X, Y, torch.randn(3, M, 3, generator=g, dtype=f64).cuda())
    # the unfused derivative solver on fp32 increments of several full-wave bands (a user-defined static kernel takes it at any size)
    for d in (0, 1):
        sk = sigkernel_amd.SigKernel(_Poly(), d)
        X, Y = walk(g, 2, 300, 3, f32), walk(g, 2, 520, 3, f32)
        sk.compute_kernel_and_derivatives_Gram(X, Y, torch.randn(2, 300, 3, generator=g).to(f32).cuda())
    # full bands: where the multi-band forward is the default for 9..16 staged fp64 dims of the RBF kernel (sweep efficiency >= 0.85)
    for d, D in itertools.product((0, 1, 2), (9, 16)):
        sk = sigkernel_amd.SigKernel(RBF(0.8), d)
        X, Y = walk(g, 3, 256, D, f64), walk(g, 4, 200, D, f64)
        sk.compute_Gram(X, Y); sk.compute_kernel(X, walk(g, 3, 200, D, f64))
    # an exported entry point the host layer has no call of any more (the one-launch loss route carries its weights from the forward)
    be.loss_weights(5, 7, torch.ones((), dtype=f64).cuda(), torch.device("cuda"))


if __name__ == "__main__":
    out = None
    if "--out" in sys.argv:
        i = sys.argv.index("--out"); out = sys.argv[i + 1]; del sys.argv[i:i + 2]
    first, stride = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 1)
    counts = run(first, stride)
    if out:
        with open(out, "w") as f:
            for name in sorted(counts):
                f.write("%d\t%s\n" % (counts[name], name))
