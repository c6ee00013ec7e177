#!/usr/bin/env python3
"""One public call, repeated -- run under `rocprofv3 --kernel-trace --stats` to see which kernels a step is made of (the glue around the
solver kernels: staging, folds, rescue scans, torch copies).  usage: r06_api_profile.py <case>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigkernel_amd
g = torch.Generator().manual_seed(0)
def walk(A, M, D, dt=torch.float64): return (torch.cumsum(torch.randn(A, M, D, generator=g, dtype=torch.float64), 1) / np.sqrt(M * D)).to(dt).cuda()
case = sys.argv[1]
LIN, RBF = sigkernel_amd.LinearKernel, sigkernel_amd.RBFKernel
f32 = torch.float32
if case == "gram_bwd_128":
    sk = sigkernel_amd.SigKernel(RBF(1.0), 1); X, Y = walk(128, 64, 3), walk(128, 64, 3)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "gram_bwd_1024_lin":
    sk = sigkernel_amd.SigKernel(LIN(), 1); X, Y = walk(1024, 64, 8), walk(1024, 64, 8)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "sym_bwd_1024":
    sk = sigkernel_amd.SigKernel(RBF(1.0), 1); X = walk(1024, 64, 3)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Xg, sym=True).sum().backward()
elif case == "mmd_512":
    sk = sigkernel_amd.SigKernel(RBF(1.0), 1); X, Y = walk(512, 64, 3), walk(512, 64, 3)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_mmd(Xg, Y).backward()
elif case == "gram_bwd_f32":
    sk = sigkernel_amd.SigKernel(RBF(1.0), 1); X, Y = walk(512, 64, 3, f32), walk(512, 64, 3, f32)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "stream_dim20":
    sk = sigkernel_amd.SigKernel(RBF(1.0), 1); X, Y = walk(256, 64, 20), walk(256, 64, 20)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "mb_bwd_300":
    sk = sigkernel_amd.SigKernel(LIN(), 1); X, Y = walk(128, 300, 8), walk(128, 300, 8)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "deriv":
    sk = sigkernel_amd.SigKernel(RBF(1.0), 1); X, Y = walk(256, 128, 8), walk(256, 128, 8); G = torch.randn(256, 128, 8, generator=g, dtype=torch.float64).cuda()
    def step(): sk.compute_kernel_and_derivatives_Gram(X, Y, G)
elif case == "swap_bwd":
    sk = sigkernel_amd.SigKernel(LIN(), 1); X, Y = walk(128, 512, 8), walk(128, 64, 8)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "lin_dim20":
    sk = sigkernel_amd.SigKernel(LIN(), 1); X, Y = walk(256, 64, 20), walk(256, 64, 20)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "rbf_dim12":
    sk = sigkernel_amd.SigKernel(RBF(1.0), 1); X, Y = walk(256, 64, 12), walk(256, 64, 12)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "generic":
    class Poly:
        def Gram_matrix(self, X, Y): return (1.0 + torch.einsum("amd,bnd->abmn", X, Y)) ** 2
        def batch_kernel(self, X, Y): return (1.0 + torch.einsum("amd,and->amn", X, Y)) ** 2
    sk = sigkernel_amd.SigKernel(Poly(), 1); X, Y = walk(128, 64, 5), walk(128, 64, 5)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "rbf_d3":
    sk = sigkernel_amd.SigKernel(RBF(1.0), 3); X, Y = walk(128, 64, 3), walk(128, 64, 3)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "c5_grad":
    sk = sigkernel_amd.SigKernel(RBF(1.0), 2); X, Y = walk(256, 512, 16, f32), walk(256, 512, 16, f32)
    def step(): Xg = X.clone().requires_grad_(True); sk.compute_Gram(Xg, Y).sum().backward()
elif case == "sym_wide":
    sk = sigkernel_amd.SigKernel(RBF(1.0), 1); X = walk(512, 64, 20)
    def step(): sk.compute_Gram(X, X, sym=True)
elif case == "deriv_generic":
    class Poly:
        def Gram_matrix(self, X, Y): return (1.0 + torch.einsum("amd,bnd->abmn", X, Y)) ** 2
        def batch_kernel(self, X, Y): return (1.0 + torch.einsum("amd,and->amn", X, Y)) ** 2
    sk = sigkernel_amd.SigKernel(Poly(), 1); X, Y = walk(128, 64, 5), walk(128, 64, 5); G = torch.randn(128, 64, 5, generator=g, dtype=torch.float64).cuda()
    def step(): sk.compute_kernel_and_derivatives_Gram(X, Y, G)
for _ in range(8): step()
torch.cuda.synchronize()
