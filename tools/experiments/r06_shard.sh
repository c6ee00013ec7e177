#!/bin/bash
# (GPU box) a 64-row shard of the headline Gram (one of 8 ranks): static share of the work queue / shares by wave age rank -> gpurun_out/r06_shard.txt
R=gpurun_out/r06_shard.txt; : > $R
for cfg in shard64 shard128; do
  for q in 0 20 35 50 75 100; do
    echo -n "SK_FUSED_Q_STATIC=$q " >> $R; SK_FUSED_Q_STATIC=$q python tools/ab.py --one new $cfg 2>&1 | grep -v amdgpu.ids >> $R
  done
  for w in "50,50" "55,45" "60,40" "66,34"; do
    echo -n "SK_FUSED_Q_STATIC=100 SK_FUSED_RANK_W=$w " >> $R; SK_FUSED_Q_STATIC=100 SK_FUSED_RANK_W=$w python tools/ab.py --one new $cfg 2>&1 | grep -v amdgpu.ids >> $R
  done
  echo -n "SK_FUSED_Q_STATIC=100 SK_FUSED_MID=0 " >> $R; SK_FUSED_Q_STATIC=100 SK_FUSED_MID=0 python tools/ab.py --one new $cfg 2>&1 | grep -v amdgpu.ids >> $R
done
cat $R
