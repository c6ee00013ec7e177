#!/bin/bash
# the kind-dependent default in place (c3 / c4fwd / c4 at the default), then the multi-band kernels' static shares (50 % by default)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2; do
  for cfg in c3 c4fwd c4 f:rbf:512:64:64:4:2; do echo -n "default  "; python tools/ab.py --one new $cfg 2>&1 | grep median; done
  for pct in 50 35 20 8; do
    echo -n "fusedmb pct $pct  "; SK_FUSEDMB_Q_STATIC=$pct python tools/ab.py --one new c5 2>&1 | grep median
    echo -n "fusedmb pct $pct  "; SK_FUSEDMB_Q_STATIC=$pct python tools/ab.py --one new f:lin:256:512:512:8:1 2>&1 | grep median
    echo -n "adjmb pct $pct  "; SK_ADJMB_Q_STATIC=$pct python tools/ab.py --one new g:rbf:128:512:512:8:1 2>&1 | grep median
    echo -n "adjmb pct $pct  "; SK_ADJMB_Q_STATIC=$pct python tools/ab.py --one new g:lin:256:512:512:8:1 2>&1 | grep median
  done
done
