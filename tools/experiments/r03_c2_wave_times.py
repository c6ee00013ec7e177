import os, sys, struct, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sigkernel_amd
gen = torch.Generator().manual_seed(0)
X = (torch.cumsum(torch.randn(128, 64, 3, generator=gen, dtype=torch.float64), 1) / np.sqrt(64 * 3)).cuda()
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
for _ in range(5): sk.compute_Gram(X, X, sym=True)
torch.cuda.synchronize()
def analyse(tag):
    raw = open(os.environ["SK_DBG_TS"], "rb").read()
    w = struct.unpack("i", raw[:4])[0]
    a = np.frombuffer(raw[4:], dtype=np.uint64).reshape(w, 6).astype(np.int64)
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0   # 100 MHz wall clock -> us
    hw, xcc, steps = a[:, 3], a[:, 4] & 0xf, a[:, 5]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
    print(tag, "waves", w, "start us: min %.1f med %.1f max %.1f | end us: min %.1f med %.1f max %.1f" % (st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max()))
    print("  duration us by steps:", {int(s): (round(float(np.median((en - st)[steps == s])), 1), int((steps == s).sum())) for s in np.unique(steps)})
    cyc = a[:, 2]
    print("  cycles/step median: %.0f" % np.median(cyc / steps), " clock GHz ~ %.2f" % np.median(cyc / ((en - st) * 1000 + 1e-9)))
    uniq, cnt = np.unique(key, return_counts=True)
    print("  distinct SIMD slots %d, waves per SIMD histogram:" % len(uniq), dict(zip(*np.unique(cnt, return_counts=True))))
    cukey = key // 10
    u2, c2 = np.unique(cukey, return_counts=True)
    print("  distinct CUs %d, waves per CU histogram:" % len(u2), dict(zip(*np.unique(c2, return_counts=True))))
    # total steps per SIMD vs last end on that SIMD
    tot = {k: steps[key == k].sum() for k in uniq}
    last = {k: en[key == k].max() for k in uniq}
    ts = np.array([tot[k] for k in uniq]); le = np.array([last[k] for k in uniq])
    for v in np.unique(ts): print("   SIMD with %d wave-steps: n=%d, last end median %.1f us max %.1f" % (v, (ts == v).sum(), np.median(le[ts == v]), le[ts == v].max()))
sk.compute_Gram(X, X, sym=True); torch.cuda.synchronize(); analyse("wpc default")
