#!/usr/bin/env python3
"""Same-box A/B of library BUILDS (same Python package): C4-shaped compute_Gram(X, Y) with a gradient -- fused RBF forward with
edges + fused RBF adjoint -- through each .so given.   usage: r03_variant.py lib.so [lib.so ...]   (alternating, 3 rounds)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import sigkernel_amd
    from sigkernel_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
    gen = torch.Generator().manual_seed(0)
    def walk(A, M, D): return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)).cuda()
    A = int(os.environ.get("SK_A", "1024"))
    X, Y = walk(A, 64, 4), walk(2048, 64, 4)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2)
    w = torch.randn(A, 2048, generator=gen, dtype=torch.float64).cuda()
    fw, bw = [], []
    for it in range(7):
        Xg = X.clone().requires_grad_(True)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); K = sk.compute_Gram(Xg, Y); e1.record(); (K * w).sum().backward(); e2.record(); torch.cuda.synchronize()
        if it >= 2: fw.append(e0.elapsed_time(e1)); bw.append(e1.elapsed_time(e2))
    print("%-28s fwd %.3f ms  bwd %.3f ms  grad checksum %r" % (os.path.basename(sys.argv[2]), float(np.median(fw)), float(np.median(bw)), float(Xg.grad.sum())), flush=True)
    sys.exit(0)
for rnd in range(3):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one", lib])
