#!/bin/bash
# is it the pairs per wave or the static kernel that decides the best static share?  (forward only: f:kind:A:M:N:D:d)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2; do
  for pct in 35 20 10; do
    for cfg in f:lin:2048:64:64:8:1 f:lin:1024:128:128:8:1 f:rbf:512:64:64:4:2 f:rbf:1024:64:64:4:2 f:rbf:1024:64:64:4:1 f:lin:512:64:64:8:1; do
      echo -n "pct $pct  "; SK_FUSED_Q_STATIC=$pct python tools/ab.py --one new $cfg 2>&1 | grep median
    done
  done
done
