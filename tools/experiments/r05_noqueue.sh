#!/bin/bash
# how much balance do static shares by wave age rank lose against the work queue on big launches?  (SK_FUSED_Q_STATIC=100: no queue)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2; do
  for cfg in e:rbf:1024:64:64:4:2 f:rbf:1024:64:64:4:2 e:lin:512:128:128:8:1 c3 e:rbf:512:128:128:4:1; do
    echo -n "queue    "; python tools/ab.py --one new $cfg 2>&1 | grep median
    echo -n "static   "; SK_FUSED_Q_STATIC=100 python tools/ab.py --one new $cfg 2>&1 | grep median
    echo -n "static, equal shares   "; SK_FUSED_MID=0 SK_FUSED_Q_STATIC=100 python tools/ab.py --one new $cfg 2>&1 | grep median
  done
done
