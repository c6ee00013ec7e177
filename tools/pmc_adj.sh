#!/bin/bash
# usage: tools/pmc_adj.sh <tag> [tune_adj args]
TAG=$1; shift
OUT=$PWD/gpurun_out/pmca_$TAG; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- python $REPO/tools/tune_adj.py "$@" > /dev/null 2> "$OUT/trace.err"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_adj_wave" -f csv -d "$OUT/$name" -o pmc -- python $REPO/tools/tune_adj.py "$@" > /dev/null 2> "$OUT/$name.err"
done
cd $REPO
python - "$OUT" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
for f in glob.glob(os.path.join(sys.argv[1],"trace","**","*kernel_stats.csv"),recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]: print(r["Name"][:90], r["Calls"], r["AverageNs"])
acc=defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-28s n=%d avg=%.5g"%(k,len(acc[k]),sum(acc[k])/len(acc[k])))
PY
