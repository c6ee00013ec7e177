#!/bin/bash
# C5's shape with a gradient: times (fused multi-band adjoint vs the unfused route) and a kernel trace of the fused route
OUT=$PWD/gpurun_out/r03g; mkdir -p $OUT; REPO=$PWD
timeout 900 python tools/r03_c5grad_time.py > $OUT/time_f32.txt 2>&1
timeout 900 python tools/r03_c5grad_time.py 256 512 16 f64 > $OUT/time_f64.txt 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python $REPO/tools/r03_c5grad_trace.py > $OUT/trace.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r03g/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print("  %-70s calls %5s avg %12.3f ms  %6s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
cat $OUT/time_f32.txt $OUT/time_f64.txt | grep -v amdgpu.ids
