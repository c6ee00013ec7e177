#!/bin/bash
# usage: tools/pmc_any.sh <kernel-regex> <script> [args]  -- memory + SQ counters for one kernel
RE=$1; shift
OUT=$PWD/gpurun_out/pmc_any; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$RE" -f csv -d "$OUT/$name" -o pmc -- python $REPO/"$@" > /dev/null 2> "$OUT/$name.err"
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
