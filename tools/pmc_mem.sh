#!/bin/bash
# memory-side counters only (fast): usage tools/pmc_mem.sh <tag> [run_fwd args]
TAG=$1; shift
OUT=$PWD/gpurun_out/pmcm_$TAG; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_fwd_wave" -f csv -d "$OUT/$name" -o pmc -- python $REPO/tools/run_fwd.py "$@" > /dev/null 2> "$OUT/$name.err"
done
cd $REPO
python - "$OUT" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  ".join("%s=%.4g"%(k.replace("TCC_EA0_","").replace("_sum",""),sum(acc[k])/len(acc[k])) for k in sorted(acc)))
PY
