#!/bin/bash
# (GPU box) the work queue's static share and the resident waves of the four-row headline kernel -> gpurun_out/r06_qstatic.txt
R=gpurun_out/r06_qstatic.txt; : > $R
for q in 0 3 20 35 50 65 100; do
  echo -n "SK_FUSED_Q_STATIC=$q " >> $R; SK_FUSED_Q_STATIC=$q python tools/ab.py --one new c3 2>&1 | grep -v amdgpu.ids >> $R
done
for q in 3 20 35 50; do
  echo -n "SK_FUSED_Q_STATIC=$q " >> $R; SK_FUSED_Q_STATIC=$q python tools/ab.py --one new "e:linear:512:128:128:8:1" 2>&1 | grep -v amdgpu.ids >> $R
  echo -n "SK_FUSED_Q_STATIC=$q " >> $R; SK_FUSED_Q_STATIC=$q python tools/ab.py --one new c2big 2>&1 | grep -v amdgpu.ids >> $R
  echo -n "SK_FUSED_Q_STATIC=$q " >> $R; SK_FUSED_Q_STATIC=$q python tools/ab.py --one new "e:rbf:1024:64:64:3:1" 2>&1 | grep -v amdgpu.ids >> $R
done
for w in 4 8; do
  echo -n "SK_FUSED_WPC=$w " >> $R; SK_FUSED_WPC=$w python tools/ab.py --one new c3 2>&1 | grep -v amdgpu.ids >> $R
done
for w in 1 2 4; do
  echo -n "SK_FUSED_WPB=$w " >> $R; SK_FUSED_WPB=$w python tools/ab.py --one new c3 2>&1 | grep -v amdgpu.ids >> $R
done
cat $R
