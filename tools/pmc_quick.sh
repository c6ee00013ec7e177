#!/bin/bash
# usage: tools/pmc_quick.sh <kernel-regex> <script> [args]  -- SQ activity counters only
RE=$1; shift
OUT=$PWD/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_EXP_GDS"; do
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
print("  ".join("%s=%.4g"%(k.replace("SQ_",""),sum(acc[k])/len(acc[k])) for k in sorted(acc)))
PY
