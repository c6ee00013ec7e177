#!/bin/bash
# kernel timelines of the small steps (one rocprofv3 --kernel-trace pass each) -> gpurun_out/r05_timelines.txt
REPO=$PWD; OUT=$REPO/gpurun_out/r05_tl; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
R=$REPO/gpurun_out/${1:-r05_timelines}.txt; : > $R
run() {  # tag first-kernel args...
  tag=$1; key=$2; shift 2
  python $REPO/tools/experiments/r05_small_steps.py "$@" >> $R 2>&1
  rm -rf $OUT/$tag
  rocprofv3 --kernel-trace -f csv -d $OUT/$tag -o t -- python $REPO/tools/experiments/r05_small_steps.py "$@" > /dev/null 2>&1
  echo "== $tag ==" >> $R
  python $REPO/tools/step_timeline.py $OUT/$tag "$key" >> $R 2>&1
}
run mmd32 k_prep_cat mmd 32
run mmd64 k_prep_cat mmd 64
run mmd128 k_prep_cat mmd 128
run c2 k_prep c2
run shard64 k_prep_pair shard 64
run shard128 k_prep_pair shard 128
cat $R
