#!/usr/bin/env python3
"""(GPU box, ONE GPU) Per-rank compute time of a strong-scaled BASELINE config for R = 1, 2, 4, 8 ranks -- c4 (default): configs[3],
compute_mmd(X, Y).backward(), 2048 x 2048 paths of length 64, dim 4, RBF, dyadic 2;  c3: configs[2], the headline,
compute_Gram of 512 x 512 paths of length 128, dim 8, LinearKernel, dyadic 1 (64 rows per rank at R = 8) --: every rank's share is run through the PRODUCT code
(sigkernel_amd.distributed: row shard of K_XY, folded triangular blocks of K_XX with the all-reduced gradient, folded K_YY)
one after the other on this GPU, with the collectives replaced by local copies of the same size.  No scaling curve can be
measured on a 1-GPU lease; what this gives is max_r t_r(R), i.e. the speed-up the kernels and the host layer allow before
communication (three all-gathers of <= 4 MB + one all-reduce of 4 MB per step, timed separately at world 1 over RCCL).

    python tools/shard_times.py [c4|c3] [out.json]
"""
import json, os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd
from sigkernel_amd import distributed as D

CFG = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("c3", "c4") else "c4"
OUT = [a for a in sys.argv[1:] if a not in ("c3", "c4")]
A, M, Dm, d = (2048, 64, 4, 2) if CFG == "c4" else (512, 128, 8, 1)
kern = (lambda: sigkernel_amd.RBFKernel(1.0)) if CFG == "c4" else (lambda: sigkernel_amd.LinearKernel())
g = torch.Generator().manual_seed(0)
mk = lambda: (torch.cumsum(torch.randn(A, M, Dm, generator=g, dtype=torch.float64), 1) / np.sqrt(M * Dm)).cuda()
X, Y = mk(), mk()

state = {"rank": 0, "world": 1}
real_dist = D.dist
shim = types.SimpleNamespace(
    get_world_size=lambda group=None: state["world"], get_rank=lambda group=None: state["rank"],
    is_initialized=lambda: True, get_backend=lambda group=None: "shim", ReduceOp=real_dist.ReduceOp, group=real_dist.group)


def fake_gather(out, inp, group):          # every rank's slot receives THIS rank's block: same bytes moved, no peer
    n = inp.shape[0]
    for r in range(state["world"]):
        out[r * n:(r + 1) * n].copy_(inp)


D.dist = shim
D._gather = fake_gather
D._all_reduce_sum = lambda t, group: t
sk_dist = sigkernel_amd.SigKernel(kern(), d, process_group="shim")
sk_one = sigkernel_amd.SigKernel(kern(), d)       # R = 1: what bench.py --gpus 1 runs (no process group)


def step():
    sk = sk_one if state["world"] == 1 else sk_dist
    if CFG == "c3":
        return sk.compute_Gram(X, Y)
    Xg = X.detach().requires_grad_(True)
    sk.compute_mmd(Xg, Y).backward()
    return Xg.grad


res = {}
for R in (1, 2, 4, 8):
    state["world"] = R
    times = []
    for r in range(R):
        state["rank"] = r
        nrep = 3 if CFG == "c4" else 30
        for _ in range(2 if CFG == "c4" else 5):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(nrep):
            step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / nrep * 1e3)
    res[R] = times
    print("R=%d: per-rank ms %s  max %.1f" % (R, ["%.1f" % t for t in times], max(times)), flush=True)
t1 = max(res[1])
out = {"workload": ("BASELINE configs[3], strong scaling: compute_mmd(X, Y).backward(), 2048 x 2048 paths, len 64, dim 4, RBF, dyadic 2, fp64" if CFG == "c4"
                    else "BASELINE configs[2] (headline), strong scaling: compute_Gram, 512 x 512 paths, len 128, dim 8, LinearKernel, dyadic 1, fp64"),
       "method": "each rank's share through sigkernel_amd.distributed on ONE MI355X, collectives replaced by local copies (tools/shard_times.py); "
                 "NOT a measured scaling curve",
       "per_rank_ms": {str(R): res[R] for R in res},
       "predicted_speedup_before_communication": {str(R): t1 / max(res[R]) for R in res}}
# the collectives of one step at world 1 over RCCL: sizes as at R = 8
try:
    D.dist = real_dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
    real_dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    blk = torch.zeros(A // 8, A, dtype=torch.float64, device="cuda"); full = torch.zeros(A // 8, A, dtype=torch.float64, device="cuda")
    gr = torch.zeros(A, M, Dm, dtype=torch.float64, device="cuda")
    for _ in range(3):
        real_dist.all_gather_into_tensor(full, blk); real_dist.all_reduce(gr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        real_dist.all_gather_into_tensor(full, blk)
        if CFG == "c4":
            real_dist.all_gather_into_tensor(full, blk); real_dist.all_gather_into_tensor(full, blk)
            real_dist.all_reduce(gr)
    torch.cuda.synchronize()
    out["collectives_ms_world1_rccl"] = (time.perf_counter() - t0) / 20 * 1e3
    real_dist.destroy_process_group()
except Exception as e:      # noqa: BLE001
    out["collectives_ms_world1_rccl"] = "failed: %s" % e
print(json.dumps(out))
if OUT:
    json.dump(out, open(OUT[0], "w"), indent=1)
