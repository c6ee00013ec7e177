#!/bin/bash
# C5's shape WITH a gradient (compute_Gram(X, Y) * w).sum().backward(): times (multi-band fused adjoint vs the unfused route, fp32 and
# fp64 tensors), a kernel trace of the fused route, and PMC counters of the two kernels.  Writes gpurun_out/r03g/.
OUT=$PWD/gpurun_out/r03g; mkdir -p $OUT; REPO=$PWD
timeout 900 python tools/r03_c5grad_time.py > $OUT/time_f32.txt 2>&1
timeout 900 python tools/r03_c5grad_time.py 256 512 16 f64 > $OUT/time_f64.txt 2>&1
timeout 900 python tools/r03_c5grad_time.py 256 512 8 f64 > $OUT/time_dim8_f64.txt 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python $REPO/tools/r03_c5grad_trace.py > $OUT/trace.log 2>&1
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  timeout 900 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_adj_fused_rbf_mb|k_fwd_fused_mb" -f csv -d "$OUT/pmc/$name" -o pmc -- python $REPO/tools/r03_c5grad_trace.py > /dev/null 2> "$OUT/pmc_$name.err"
done
cd $REPO
python - <<'PY' > gpurun_out/r03g/summary.txt
import csv, glob, os
from collections import defaultdict
print("== compute_Gram(X, Y) with a gradient at BASELINE configs[4]'s shape: rocprofv3 --kernel-trace --stats (tools/r03_c5grad_trace.py, 2 steps) ==")
for f in glob.glob("gpurun_out/r03g/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("  %-72s calls %5s avg %12.3f ms  %6s %%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
print("== PMC, averages per dispatch ==")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/r03g/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(" ", k)
    for c in sorted(acc[k]): print("      %-28s n=%d avg=%.6g" % (c, len(acc[k][c]), sum(acc[k][c]) / len(acc[k][c])))
for t in ("time_f32", "time_f64", "time_dim8_f64"):
    print("==", t, "==")
    print("".join(l for l in open("gpurun_out/r03g/%s.txt" % t) if "amdgpu.ids" not in l), end="")
PY
cat gpurun_out/r03g/summary.txt
