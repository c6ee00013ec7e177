#!/bin/bash
# NOTE: SK_FUSED_RC4 existed only in the experiment build of round 6 (it chose the rows per lane at run time); the rule is now fused_rcx in
# csrc/sk_wave_fused.hip and the knob is gone -- this script documents how profiles/r06_rc4_ab.txt was produced.
# (GPU box) four coarse rows per lane at dyadic 1 (SK_FUSED_RC4=2: wherever in scope) against two (=0), same box, alternating:
# the BASELINE-derived shapes the forward kernel serves -> gpurun_out/r06_rc4_ab.txt
R=gpurun_out/${1:-r06_rc4_ab}.txt; : > $R
for cfg in c3 shard64 shard128 c2 c2big mmd32 mmd64 mmd128 "f:rbf:512:33:33:3:1" "e:rbf:1024:64:64:3:1" "f:linear:1024:64:64:4:1" \
           "e:linear:512:128:128:8:1" "g:linear:512:128:128:8:1" "f:linear:512:100:128:8:1" "f:rbf:512:128:128:3:1"; do
  for rnd in 1 2; do
    for rc in 0 2; do
      echo -n "RC4=$rc " >> $R
      SK_FUSED_RC4=$rc python tools/ab.py --one new "$cfg" 2>&1 | grep -v amdgpu.ids >> $R
    done
  done
done
cat $R
