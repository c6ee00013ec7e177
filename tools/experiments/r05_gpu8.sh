cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; R=$PWD; cd /tmp
for which in new r04; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
    rm -rf /tmp/pq; rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_fwd_fused" -f csv -d /tmp/pq -o pmc -- python $R/tools/ab.py --one $which c3 > /dev/null 2>&1
    python - $which <<'PY'
import csv,glob,sys
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("/tmp/pq/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: "%.5g"%(sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
  done
done
