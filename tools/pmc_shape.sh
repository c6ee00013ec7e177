#!/bin/bash
# (GPU box) PMC passes of one compute_Gram shape, restricted to the fused forward kernels: pmc_shape.sh <tag> kind A M N D d
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-30)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_fwd_fused" -f csv -d "$OUT/$name" -o pmc -- python $REPO/tools/experiments/r04_one_shape.py "$@" > /dev/null 2> "$OUT/$name.err" || echo "failed: $set"
done
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
for c in sorted(acc):
    v = sorted(acc[c]); print("  %-26s n=%d  median %.5g" % (c, len(v), v[len(v) // 2]))
PY
