import os, sys, struct, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sigkernel_amd
gen = torch.Generator().manual_seed(0)
walk = lambda A, M, D: (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), 1) / np.sqrt(M * D)).cuda()
X, Y = walk(512, 128, 8), walk(512, 128, 8)
sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1)
for _ in range(12): sk.compute_Gram(X, Y)
torch.cuda.synchronize()
for rep in range(2):
    sk.compute_Gram(X, Y); torch.cuda.synchronize()
    raw = open(os.environ["SK_DBG_TS"], "rb").read()
    w = struct.unpack("i", raw[:4])[0]
    a = np.frombuffer(raw[4:], dtype=np.uint64).reshape(w, 6).astype(np.int64)
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    steps = a[:, 5]
    print("waves", w, "start us: med %.1f max %.1f | end us: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (np.median(st), st.max(), en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
    print("  steps per wave: min %d med %d max %d ; total wave-steps %d ; us per step (per wave) median %.3f" % (steps.min(), np.median(steps), steps.max(), steps.sum(), np.median((en - st) / steps)))
    hw, xcc = a[:, 3], a[:, 4] & 0xf
    for x in range(8):
        m = xcc == x
        print("   xcc %d: waves %d, end med %.1f max %.1f, steps sum %d" % (x, m.sum(), np.median(en[m]), en[m].max(), steps[m].sum()))
