#!/bin/bash
# A round's profiles (GPU box): rocprofv3 kernel trace + stats of bench.py on c3 / c5 / c4, then separate PMC passes
# (counters never combined with API tracing) restricted to the dominant kernels.  usage: profile_round.sh <tag>  ->  gpurun_out/<tag>p/
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/${TAG}p; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
cd /tmp
for cfg in c3 c5 c4; do
  steps=5; warm=1; [ $cfg = c4 ] && steps=2; [ $cfg = c3 ] && steps=40 && warm=10   # (short launches: the clocks take ~10 launches to settle)
  timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$cfg -o trace -- python $REPO/bench.py --config $cfg --steps $steps --warmup $warm --no-extras > $OUT/trace_$cfg.json 2> $OUT/trace_$cfg.err
done
pmc() {  # tag regex bench-args...
  tag=$1; re=$2; shift 2
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE"; do
    name=$(echo $set | tr ' ' '+' | cut -c1-30)
    timeout 600 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$re" -f csv -d "$OUT/pmc_$tag/$name" -o pmc -- python $REPO/bench.py "$@" > /dev/null 2> "$OUT/pmc_$tag/$name.err" || echo "failed: $tag $set" >> $OUT/failed.txt
  done
}
mkdir -p $OUT/pmc_c5 $OUT/pmc_c4 $OUT/pmc_c3 $OUT/pmc_c2
pmc c2 "k_fwd_fused" --config c2 --steps 3 --warmup 1 --no-extras
pmc c5 "k_fwd_fused_mb" --config c5 --steps 2 --warmup 1 --no-extras
pmc c4 "k_adj_fused|k_fwd_fused|k_fused_rescue|k_screen" --config c4 --steps 1 --warmup 1 --no-extras
pmc c3 "k_fwd_fused" --config c3 --steps 3 --warmup 1 --no-extras
cd $REPO
python - $OUT <<'PY' > $OUT/summary.txt 2>&1
import csv,glob,os,sys
from collections import defaultdict
out=sys.argv[1]
def short(n):
    n=n.replace("void sk::(anonymous namespace)::","")
    return n.split("(")[0][:64]
for cfg in ("c3","c5","c4"):
    f=glob.glob(os.path.join(out,"trace_"+cfg,"**","*kernel_stats.csv"),recursive=True)
    print("== %s: rocprofv3 --kernel-trace --stats (bench.py --config %s --no-extras) =="%(cfg,cfg))
    if f:
        for r in list(csv.DictReader(open(f[0])))[:12]:
            print("  %-66s calls %5s  total %12s ns  avg %12s ns  %6s %%"%(short(r["Name"]),r["Calls"],r["TotalDurationNs"],r["AverageNs"],r["Percentage"]))
    try: print("  bench line:", open(os.path.join(out,"trace_%s.json"%cfg)).read()[:400])
    except Exception: pass
    kt=glob.glob(os.path.join(out,"trace_"+cfg,"**","*kernel_trace.csv"),recursive=True)
    if kt and f:
        top=list(csv.DictReader(open(f[0])))[0]["Name"]
        d=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open(kt[0])) if r["Kernel_Name"]==top)
        print("  per dispatch of the top kernel (ms): first-to-last sorted min %.3f  median %.3f  max %.3f  (n=%d; the first launches of a process run at lower clocks)"%(d[0],d[len(d)//2],d[-1],len(d)))
for cfg in ("c5","c4","c3","c2"):
    print("== %s: PMC, averages per dispatch and kernel =="%cfg)
    acc=defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out,"pmc_"+cfg,"**","*counter_collection.csv"),recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(" ",k)
        for c in sorted(acc[k]):
            v=acc[k][c]; print("      %-28s n=%-4d avg=%.6g"%(c,len(v),sum(v)/len(v)))
if os.path.exists(os.path.join(out,"failed.txt")): print(open(os.path.join(out,"failed.txt")).read())
PY
cat $OUT/summary.txt | head -150
# the HBM-streaming solver's own PMC pass, the per-rank shard times, this round's bench lines, traffic.json from THESE counters
bash tools/pmc_fwd.sh ${TAG}fwd > $OUT/pmc_fwd_solver.txt 2>&1
python tools/traffic.py $OUT gpurun_out/pmc_${TAG}fwd $TAG > $OUT/traffic.log 2>&1
timeout 900 python tools/shard_times.py c4 $OUT/c4_shard_times.json > $OUT/shard_c4.log 2>&1
timeout 900 python tools/shard_times.py c3 $OUT/c3_shard_times.json > $OUT/shard_c3.log 2>&1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for cfg in c2 c4 c5; do timeout 900 python bench.py --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; done
