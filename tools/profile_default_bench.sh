#!/bin/bash
# (GPU box) rocprofv3 --kernel-trace --stats of the DRIVER's bench command, and the same command's JSON line without the profiler:
# the kernel's average duration in the stats must agree with roofline.avg_launch_ms of the line.  usage: profile_default_bench.sh <tag>
set -u
TAG=${1:-r04}; OUT=$PWD/gpurun_out/${TAG}d; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-live-traffic --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err
cd $REPO
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv
python - $OUT <<'PY'
import csv, json, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
line = json.loads([l for l in open(out + "/bench_line.json") if l.startswith("{")][-1])
print("bench line: ms_per_step %.4f, roofline.avg_launch_ms %.4f, frac %.4f, traffic %.4g B (%s)" % (
    line["ms_per_step"], line["roofline"]["avg_launch_ms"], line["roofline"]["frac"], line["roofline"]["traffic"], line["roofline"]["traffic_source"][:40]))
for r in rows[:8]:
    print("  %-90s calls %5s avg %12.1f ns  %6s %%" % (r["Name"].replace("void sk::(anonymous namespace)::", "")[:90], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
