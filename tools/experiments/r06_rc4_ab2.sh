#!/bin/bash
# NOTE: SK_FUSED_RC4 existed only in the experiment build of round 6 (it chose the rows per lane at run time); the rule is now fused_rcx in
# csrc/sk_wave_fused.hip and the knob is gone -- this script documents how profiles/r06_rc4_ab.txt was produced.
# (GPU box) dyadic 2: two coarse rows per lane (SK_FUSED_RC4=6) against one (=2), same box, alternating -> gpurun_out/r06_rc4_ab2.txt
R=gpurun_out/${1:-r06_rc4_ab2}.txt; : > $R
python tools/experiments/r06_rc4.py --parity-only 2>&1 | grep -v amdgpu.ids >> $R
for cfg in c4fwd "e:rbf:1024:64:64:4:2" "f:linear:512:64:64:8:2" "e:linear:512:64:64:8:2" "g:rbf:512:64:64:4:2" "f:rbf:512:33:33:3:2" c4; do
  for rnd in 1 2; do
    for rc in 2 6; do
      echo -n "RC4=$rc " >> $R
      SK_FUSED_RC4=$rc python tools/ab.py --one new "$cfg" 2>&1 | grep -v amdgpu.ids >> $R
    done
  done
done
cat $R
