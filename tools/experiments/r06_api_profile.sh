#!/bin/bash
# (GPU box) rocprofv3 kernel summaries of a handful of public calls -> gpurun_out/r06_api_profile.txt
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/r06_api_profile.txt; : > $OUT; cd /tmp
for c in ${@:-gram_bwd_128 gram_bwd_1024_lin sym_bwd_1024 mmd_512 gram_bwd_f32 stream_dim20 mb_bwd_300 deriv swap_bwd lin_dim20 rbf_dim12 generic rbf_d3}; do
  rm -rf /tmp/apiprof; rocprofv3 --kernel-trace --stats -f csv -d /tmp/apiprof -o p -- python $R/tools/experiments/r06_api_profile.py $c > /dev/null 2>&1
  echo "== $c (8 steps)" >> $OUT
  python - >> $OUT <<PY
import csv,glob
f=glob.glob("/tmp/apiprof/**/p_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("  total kernel time per step %.3f ms" % (tot/8e6))
for r in rows[:9]: print("  %-96s calls %4s avg_us %9.1f  %5.1f %%" % (r["Name"].replace("void sk::(anonymous namespace)::","").replace("sk::(anonymous namespace)::","")[:96], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
done
cat $OUT
