#!/bin/bash
# (GPU box) Where does a launch of one or two waves per SIMD spend its cycles?  SQ counters (separate --pmc passes, kernel trace only)
# of the two solver launches of a 32+32-path compute_mmd().backward() step, of the C2 launch (two waves per SIMD) and of a 64-row shard
# of the headline (three waves per SIMD, for the saturated figure), plus the lone-wave issue microbenchmark.
#   usage: tools/experiments/r06_small_pmc.sh [tag]      -> gpurun_out/<tag>.txt   (default tag r06_small_launch_pmc)
set -u
TAG=${1:-r06_small_launch_pmc}
REPO=$PWD; OUT=$REPO/gpurun_out/${TAG}_raw; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
R=$REPO/gpurun_out/$TAG.txt; : > $R
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"
      "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH"
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_SENDMSG SQ_THREAD_CYCLES_VALU"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"
      "GRBM_GUI_ACTIVE")
run() {   # name regex workload-args...
  name=$1; re=$2; shift 2
  echo "== $name: kernels matching $re (r05_small_steps.py $*) ==" >> $R
  python $REPO/tools/experiments/r05_small_steps.py "$@" >> $R 2>&1
  i=0
  for set in "${SETS[@]}"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$re" -f csv -d "$OUT/$name/$i" -o pmc -- \
        python $REPO/tools/experiments/r05_small_steps.py "$@" 12 > /dev/null 2> "$OUT/$name.$i.err" || echo "  failed: $set" >> $R
  done
  python - "$OUT/$name" >> $R <<'PY'
import csv, glob, sys, os, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float); kn = {}
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); kn[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, c), v in per.items():
        m = re.search(r"(k_\w+<[^>]*>)", kn[d])
        acc[m.group(1) if m else kn[d][:60]][c].append(v)
for k in sorted(acc):
    print("  kernel %s" % k)
    m = {c: sorted(v)[len(v) // 2] for c, v in acc[k].items()}
    for c in sorted(m): print("    %-26s n=%-3d median %.6g" % (c, len(acc[k][c]), m[c]))
    try:
        w, wc = m["SQ_WAVES"], m["SQ_WAVE_CYCLES"]
        tot = sum(m.get(c, 0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD"))
        print("    -> per wave: %.0f VALU, %.0f SALU (of which %.0f branches), %.0f LDS, %.0f SMEM instructions; wave cycles (x4) per instruction of any kind: %.2f, per VALU: %.2f"
              % (m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m.get("SQ_INSTS_BRANCH", 0) / w, m["SQ_INSTS_LDS"] / w, m.get("SQ_INSTS_SMEM", 0) / w, 4 * wc / tot, 4 * wc / m["SQ_INSTS_VALU"]))
    except Exception as e:
        print("    (no summary: %s)" % e)
PY
}
run mmd32 "k_fwd_fused|k_adj_fused_rbf" mmd 32
run mmd64 "k_fwd_fused|k_adj_fused_rbf" mmd 64
run c2 "k_fwd_fused" c2
run shard64 "k_fwd_fused" shard 64
echo "== tools/ubench/lone_issue ==" >> $R
$REPO/tools/ubench/lone_issue >> $R 2>&1
cat $R
