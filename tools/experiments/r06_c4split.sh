#!/bin/bash
# (GPU box) per-kernel times of a BASELINE configs[3] step for builds under ab_base/ and the working tree, one box: usage r06_c4split.sh [base ...]
export TMPDIR=/tmp; R=$PWD
BUILDS="${@:-r06a} new"
for b in $BUILDS; do
  SK_AB_BASE=$b python tools/ab.py --one $b c4 2>&1 | grep -v amdgpu
done
cd /tmp
for b in $BUILDS; do
  rm -rf /tmp/c4p_$b
  rocprofv3 --kernel-trace --stats -f csv -d /tmp/c4p_$b -o c4 -- python $R/tools/ab.py --one $b c4 > /dev/null 2>&1
  echo "== $b kernel stats"; head -5 $(find /tmp/c4p_$b -name "*kernel_stats.csv" | head -1) | cut -c1-190
done
