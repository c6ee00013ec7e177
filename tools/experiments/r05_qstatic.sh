#!/bin/bash
# round 5: the static share of the work queue (SK_FUSED_Q_STATIC, 35 % by default) for launches whose forward keeps edges: the look-up
# of a drawn chunk's first pair sits in the macro-step path there, a bigger static share makes it rarer.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2; do
  for pct in ${PCTS:-35 50 65 80 92}; do
    for cfg in c4 c4fwd g:rbf:1024:64:64:4:2 c3; do
      echo -n "pct $pct  "; SK_FUSED_Q_STATIC=$pct python tools/ab.py --one new $cfg 2>&1 | grep median
    done
  done
done
