#!/bin/bash
# usage: tools/pmc_fwd.sh <tag> [run_fwd args]   (env knobs such as SK_WAVE_WPC are inherited)
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_SECTORS_sum" \
           "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-60)
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_fwd_wave" -f csv -d "$OUT/$name" -o pmc -- python $REPO/tools/run_fwd.py "$@" > /dev/null 2> "$OUT/$name.err" || echo "failed: $set" >> $OUT/failed.txt
done
cd $REPO; python tools/summarize_profile.py $OUT/.. 2>/dev/null | head -0
python - "$OUT" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-32s n=%d avg=%.6g"%(k,len(acc[k]),sum(acc[k])/len(acc[k])))
PY
