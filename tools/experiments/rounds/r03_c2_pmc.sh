#!/bin/bash
# (GPU) SQ counters of the C2 forward kernel: instructions per wave and per macro-step, busy cycles
OUT=$PWD/gpurun_out/c2pmc; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_fwd_fused" -f csv -d "$OUT/$name" -o pmc -- python $REPO/bench.py --config c2 --no-extras --steps 20 --warmup 5 > /dev/null 2> "$OUT/$name.err"
done
cd $REPO
python - "$OUT" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-28s n=%d avg=%.6g"%(k,len(acc[k]),sum(acc[k])/len(acc[k])))
PY
cd /tmp; rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python $REPO/bench.py --config c2 --no-extras --steps 50 --warmup 5 > $OUT/bench.log 2>&1
cd $REPO; python tools/summarize_profile.py $OUT/trace 2>/dev/null | head -12; tail -1 $OUT/bench.log | cut -c1-300
