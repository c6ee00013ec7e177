#!/bin/bash
# kernel trace of one config with the working tree: gpurun_out/r03t/<cfg>_kernels.txt
set -u
cfg=${1:-c4}; OUT=$PWD/gpurun_out/r03t; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$cfg -o trace -- python $REPO/bench.py --config $cfg --steps 3 --warmup 2 --no-extras > $OUT/trace_$cfg.json 2> $OUT/trace_$cfg.err
cd $REPO
python tools/r02_kstat.py $OUT/trace_$cfg k_ | tee $OUT/${cfg}_kernels.txt
python - $OUT/trace_$cfg <<'PY'
import csv,glob,re,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
def short(n): return re.sub(r"void sk::\(anonymous namespace\)::","",n).split("(")[0][:58]
for r in rows[-70:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if d>1.0: print("%9.3f ms  %s" % (d, short(r["Kernel_Name"])))
PY
