#!/usr/bin/env python3
"""Same-box A/B of an older build against the working tree: box-to-box variance is +-5 %, so only timings from ONE gpurun call
compare.  The base is a copy of the package (its .py files + libsigkernel_amd.so built from that commit) under ab_base/<name>/
(git-ignored; it travels to the GPU box): `git worktree add /tmp/base <commit> && make -C /tmp/base/sigkernel_amd/csrc`, then copy
sigkernel_amd/*.py and the .so.
usage: SK_AB_BASE=<name> ab.py [c3|c4fwd|c4|c5|c2|c2big ...]   (each build runs in its own process, alternating, three rounds)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    which, cfg = sys.argv[2], sys.argv[3]
    sys.path.insert(0, os.path.join(ROOT, "ab_base", which) if which != "new" else ROOT)
    import numpy as np, torch
    import sigkernel_amd
    gen = torch.Generator().manual_seed(0)
    def walk(A, M, D, dt=torch.float64): return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)).to(dt).cuda()
    if cfg == "c3":
        X, Y = walk(512, 128, 8), walk(512, 128, 8); sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1); fn = lambda: sk.compute_Gram(X, Y)
    elif cfg == "c2":
        X = walk(128, 64, 3); sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1); fn = lambda: sk.compute_Gram(X, X, sym=True)
    elif cfg == "c2big":
        X, Y = walk(512, 64, 3), walk(512, 64, 3); sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1); fn = lambda: sk.compute_Gram(X, Y)
    elif cfg == "c5":
        X, Y = walk(256, 512, 16, torch.float32), walk(256, 512, 16, torch.float32); sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2); fn = lambda: sk.compute_Gram(X, Y)
    elif cfg.startswith("g:") or cfg.startswith("f:") or cfg.startswith("e:"):
        # g:kind:A:M:N:D:d[:naive]  compute_Gram(X, Y) with a weighted-sum backward (f: forward only, e: forward that keeps edges) on a free shape -- what the
        # round-4 route changes are checked with (shapes that streamed increments before)
        _, kind, A, M, N, D, d = cfg.split(":")[:7]
        naive = cfg.endswith(":naive")
        A, M, N, D, d = int(A), int(M), int(N), int(D), int(d)
        X, Y = walk(A, M, D), walk(A, N, D)
        w = torch.randn(A, A, generator=gen, dtype=torch.float64).cuda()
        k = sigkernel_amd.RBFKernel(1.0) if kind == "rbf" else sigkernel_amd.LinearKernel()
        sk = sigkernel_amd.SigKernel(k, d, _naive_solver=naive)
        if cfg.startswith("f:"):
            fn = lambda: sk.compute_Gram(X, Y)
        elif cfg.startswith("e:"):      # the forward of a call with a gradient pending (keeps the terminal edges), no backward
            Xg = X.detach().requires_grad_(True)
            fn = lambda: sk.compute_Gram(Xg, Y).detach()
        else:
            def fn():
                Xg = X.detach().requires_grad_(True); (sk.compute_Gram(Xg, Y) * w).sum().backward(); return Xg.grad
    elif cfg in ("mmd32", "mmd64", "mmd128"):
        n_ = int(cfg[3:]); X, Y = walk(n_, 64, 3), walk(n_, 64, 3); sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
        def fn():
            Xg = X.detach().requires_grad_(True); sk.compute_mmd(Xg, Y).backward(); return Xg.grad
    elif cfg.startswith("shard"):      # rows of the headline Gram one of 512 / rows ranks solves
        X, Y = walk(int(cfg[5:]), 128, 8), walk(512, 128, 8); sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1); fn = lambda: sk.compute_Gram(X, Y)
    elif cfg == "c4fwd":
        X, Y = walk(2048, 64, 4), walk(2048, 64, 4); sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2); fn = lambda: sk.compute_Gram(X, Y)
    else:
        X, Y = walk(2048, 64, 4), walk(2048, 64, 4); sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2)
        def fn():
            Xg = X.detach().requires_grad_(True); sk.compute_mmd(Xg, Y).backward(); return Xg.grad
    n = 30 if cfg in ("c3", "c2", "c2big") or cfg.startswith("mmd") or cfg.startswith("shard") else 6
    if ":" in cfg: n = 10
    for _ in range(max(3, n // 3)): out = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print("%-4s %-6s median %.4f ms  min %.4f ms  checksum %r" % (which, cfg, float(np.median(ts)), min(ts), float(out.double().sum())), flush=True)
    sys.exit(0)
for cfg in sys.argv[1:] or ["c3"]:
    for rnd in range(3):
        for which in (os.environ.get("SK_AB_BASE", "base"), "new"):
            subprocess.run([sys.executable, __file__, "--one", which, cfg])
