#!/bin/bash
# usage: tools/trace_any.sh <python script + args>   -> top kernels by time
OUT=/tmp/tr_any; rm -rf $OUT; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o t -- python $REPO/"$@" > /tmp/tr_any.out 2>&1
tail -3 /tmp/tr_any.out
python - <<'PY'
import csv,glob
for f in glob.glob("/tmp/tr_any/**/*kernel_stats.csv",recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print("%-95s calls=%-4s avg_us=%-9.0f pct=%s"%(r["Name"][:95], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
