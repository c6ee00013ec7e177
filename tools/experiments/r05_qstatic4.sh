#!/bin/bash
# same-box check of the kind-dependent default (RBF 20 %) against the old 35 %
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2 3; do
  for cfg in c4fwd c4 f:rbf:512:64:64:4:2 g:rbf:1024:64:64:4:2 f:rbf:1024:64:64:4:1; do
    echo -n "old35    "; SK_FUSED_Q_STATIC=35 python tools/ab.py --one new $cfg 2>&1 | grep median
    echo -n "default  "; python tools/ab.py --one new $cfg 2>&1 | grep median
  done
done
