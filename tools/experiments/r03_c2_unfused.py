import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sigkernel_amd
gen = torch.Generator().manual_seed(0)
X = (torch.cumsum(torch.randn(128, 64, 3, generator=gen, dtype=torch.float64), 1) / np.sqrt(64 * 3)).cuda()
sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1)
sym = len(sys.argv) > 1 and sys.argv[1] == "sym"
for _ in range(30): K = sk.compute_Gram(X, X, sym=sym)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): K = sk.compute_Gram(X, X, sym=sym)
e1.record(); torch.cuda.synchronize(); print("sym" if sym else "full", os.environ.get("SK_NO_FUSED_RBF"), "%.1f us/call" % (e0.elapsed_time(e1) * 10))
