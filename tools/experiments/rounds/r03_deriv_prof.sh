#!/bin/bash
# compute_kernel_and_derivatives_Gram at the shape of profiles/r01_deriv_kernel.txt (256 x 256 pairs, len 128, dim 8, d = 1) and a longer
# one: fused derivative solver vs the unfused route -- end-to-end times and kernel traces.  Writes gpurun_out/r03d/.
OUT=$PWD/gpurun_out/r03d; mkdir -p $OUT; REPO=$PWD
export TMPDIR=/tmp
{
for k in linear rbf; do
  for shape in "256 256 128 8 1" "256 256 128 8 0" "128 128 256 4 1" "256 256 160 8 1"; do
    echo -n "fused   : "; python tools/time_kgrad.py $shape $k 2>/dev/null
    echo -n "unfused : "; SK_NO_FUSED_DERIV=1 python tools/time_kgrad.py $shape $k 2>/dev/null
  done
done
} > $OUT/times.txt 2>&1
cd /tmp
for k in linear rbf; do
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_fused_$k -o trace -- python $REPO/tools/time_kgrad.py 256 256 128 8 1 $k > /dev/null 2>&1
  SK_NO_FUSED_DERIV=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_unfused_$k -o trace -- python $REPO/tools/time_kgrad.py 256 256 128 8 1 $k > /dev/null 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-include-regex "k_deriv_fused" -f csv -d $OUT/pmc -o pmc -- python $REPO/tools/time_kgrad.py 256 256 128 8 1 linear > /dev/null 2>&1
cd $REPO
python - <<'PY' > gpurun_out/r03d/summary.txt
import csv, glob
from collections import defaultdict
print("== compute_kernel_and_derivatives_Gram, end to end (tools/time_kgrad.py A B len dim dyadic kernel) ==")
print(open("gpurun_out/r03d/times.txt").read())
for tag in ("fused_linear", "unfused_linear", "fused_rbf", "unfused_rbf"):
    print("== rocprofv3 --kernel-trace --stats, 256 x 256 pairs, len 128, dim 8, d = 1, %s (7 calls) ==" % tag)
    for f in glob.glob("gpurun_out/r03d/trace_%s/**/*kernel_stats.csv" % tag, recursive=True):
        for r in list(csv.DictReader(open(f)))[:5]:
            print("  %-72s calls %5s avg %10.3f ms  %6s %%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
acc = defaultdict(list)
for f in glob.glob("gpurun_out/r03d/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC of k_deriv_fused<1, 0, 8, LDS boundary>, averages per dispatch ==")
for c in sorted(acc): print("      %-28s n=%d avg=%.6g" % (c, len(acc[c]), sum(acc[c]) / len(acc[c])))
PY
cat gpurun_out/r03d/summary.txt
