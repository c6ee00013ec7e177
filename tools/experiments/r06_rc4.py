#!/usr/bin/env python3
"""(GPU box; ran on the round-6 EXPERIMENT build, which chose the rows per lane by SK_FUSED_RC4 -- the knob is gone, fused_rcx in
csrc/sk_wave_fused.hip is the rule) Four coarse rows per lane at dyadic 1 (k_fwd_fused<..., RCX = 4>) against two: bit-identity and the per-macro-step
costs the launcher's model (fused_small_cost, csrc/sk_wave_fused.hip) is calibrated by.
   r06_rc4.py                 drives itself: every part once per SK_FUSED_RC4 setting (a knob is read once per process)
   r06_rc4.py --dump f.npz    one process: values and gradients of the parity shapes -> f.npz
   r06_rc4.py --time          one process: forward launches of exactly k pairs per lane group at q waves per SIMD"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SHAPES = [(3, 4, 10, 20, 2), (5, 7, 33, 7, 3), (40, 50, 64, 64, 3), (17, 9, 65, 100, 4), (6, 5, 128, 31, 1), (64, 64, 20, 200, 3),
          (32, 32, 100, 64, 2)]


def walk(gen, A, M, D):
    import torch
    return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()


def dump(path):
    import torch, sigkernel_amd
    gen = torch.Generator().manual_seed(5)
    out = {}
    cases = [(sh, d, kn) for d in (1, 2) for kn in ("rbf", "linear") for sh in SHAPES if (sh[2] - 1) <= (128 >> (d - 1)) or kn == "linear"]
    for i, ((A, B, M, N, D), d, kn) in enumerate(cases):
        if (M - 1 if kn == "linear" else M) > (128 if d == 1 else 64):
            continue
        X, Y = walk(gen, A, M, D), walk(gen, B, N, D)
        w = torch.randn(A, B, generator=gen, dtype=torch.float64).cuda()
        sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.7) if kn == "rbf" else sigkernel_amd.LinearKernel(), d)
        out["K%d" % i] = sk.compute_Gram(X, Y).cpu().numpy()
        Xg = X.detach().requires_grad_(True)
        Kg = sk.compute_Gram(Xg, Y)
        (Kg * w).sum().backward()
        out["Ke%d" % i], out["g%d" % i] = Kg.detach().cpu().numpy(), Xg.grad.cpu().numpy()
        out["S%d" % i] = sk.compute_Gram(X, X, sym=True).cpu().numpy()
        if M == N:
            Xg = X.detach().requires_grad_(True)
            m = sk.compute_mmd(Xg, Y)
            m.backward()
            out["m%d" % i], out["mg%d" % i] = m.detach().cpu().numpy(), Xg.grad.cpu().numpy()
            out["p%d" % i] = sk.compute_kernel(X[:min(A, B)], Y[:min(A, B)]).cpu().numpy()
    np.savez(path, **out)


def timing():
    import torch, sigkernel_amd
    gen = torch.Generator().manual_seed(0)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
    rc4 = os.environ.get("SK_FUSED_RC4") == "2"
    G = 4 if rc4 else 2
    q = int(os.environ.get("SK_FUSED_WPC", "4")) // 4
    for M in (64, 32):
        Gm = G if M == 64 else 2 * G
        for edges in (False, True):
            for k in (1, 2, 3, 4):
                P = 1024 * q * Gm * k
                A = P // 256
                X, Y = walk(gen, A, M, 3), walk(gen, 256, M, 3)
                Xg = X.detach().requires_grad_(True)
                fn = (lambda: sk.compute_Gram(Xg, Y)) if edges else (lambda: sk.compute_Gram(X, Y))
                for _ in range(5): fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(20):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
                print("rc%d q=%d M=%3d edges=%d k=%d pairs=%6d : median %7.1f us  min %7.1f us" % (4 if rc4 else 2, q, M, edges, k, P, float(np.median(ts)), min(ts)), flush=True)


if __name__ == "__main__":
    if "--dump" in sys.argv:
        dump(sys.argv[sys.argv.index("--dump") + 1])
    elif "--time" in sys.argv:
        timing()
    else:
        tmp = os.path.join(ROOT, "gpurun_out")
        os.makedirs(tmp, exist_ok=True)
        files = {}
        for rc in ("0", "6"):
            files[rc] = os.path.join(tmp, "r06_rc4_dump_%s.npz" % rc)
            subprocess.check_call([sys.executable, __file__, "--dump", files[rc]], env=dict(os.environ, SK_FUSED_RC4=rc))
        a, b = np.load(files["0"]), np.load(files["6"])
        worst = 0.0
        for key in a.files:
            same = np.array_equal(a[key], b[key])
            d = float(np.max(np.abs(a[key] - b[key])) / max(1e-300, np.max(np.abs(a[key]))))
            worst = max(worst, d)
            if not same:
                print("  %-5s differs: max rel %.3g" % (key, d))
        print("parity: %d arrays, two rows per lane vs four: worst relative difference %.3g" % (len(a.files), worst), flush=True)
        if "--parity-only" in sys.argv:
            sys.exit(0)
        for rc in ("0", "2"):
            for wpc in ("4", "8", "12"):
                if rc == "2" and wpc == "12":
                    continue
                subprocess.call([sys.executable, __file__, "--time"], env=dict(os.environ, SK_FUSED_RC4=rc, SK_FUSED_WPC=wpc))
