#!/bin/bash
OUT=$PWD/gpurun_out/trace_adj; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT" -o trace -- python $REPO/tools/tune_adj.py "$@" > /dev/null 2> "$OUT/trace.err"
python - "$OUT" <<'PY'
import csv,glob,sys,os
for f in glob.glob(os.path.join(sys.argv[1],"**","*kernel_stats.csv"),recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]: print(r["Name"][31:110], r["Calls"], "avg_us=%.0f"%(float(r["AverageNs"])/1e3))
PY
