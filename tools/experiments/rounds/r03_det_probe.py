#!/usr/bin/env python3
"""Run-to-run determinism of the multi-band forward / adjoints per kernel variant: the same backward six times, bitwise comparison."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
def walk(g, A, M, D, dt): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt)
for kname in ("rbf", "linear"):
    for (d, A, B, M, N, D, dt) in [(1, 6, 7, 300, 200, 12, torch.float64), (2, 6, 7, 150, 200, 12, torch.float64), (2, 6, 7, 150, 200, 16, torch.float32),
                                   (1, 6, 7, 300, 200, 16, torch.float32), (0, 6, 7, 300, 200, 12, torch.float64), (1, 6, 7, 300, 200, 6, torch.float64),
                                   (2, 6, 7, 150, 200, 6, torch.float64)]:
        if kname == "rbf" and d == 0: continue
        g = torch.Generator().manual_seed(1)
        X, Y = walk(g, A, M, D, dt).cuda(), walk(g, B, N, D, dt).cuda()
        w = torch.randn(A, B, generator=g).to(dt).cuda()
        k = sigkernel_amd.RBFKernel(0.9) if kname == "rbf" else sigkernel_amd.LinearKernel()
        sk = sigkernel_amd.SigKernel(k, d)
        outs = []
        for _ in range(6):
            Xg = X.clone().requires_grad_(True)
            K = sk.compute_Gram(Xg, Y)
            (K * w).sum().backward()
            outs.append((K.detach().clone(), Xg.grad.clone()))
        nk = len({tuple(o[0].flatten().tolist()) for o in outs}); ng = len({tuple(o[1].flatten().tolist()) for o in outs})
        print(kname, (d, A, B, M, N, D, str(dt)[6:]), "distinct forward values", nk, "distinct gradients", ng, flush=True)
