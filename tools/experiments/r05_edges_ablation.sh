#!/bin/bash
# What the edge-keeping forward pays for (timing ablations; the ablated builds give WRONG edges -- ab_base/expA: no edge stores (the
# compiler then drops the bookkeeping that feeds them too), expB: no look-ups of a chunk's first pair, expAB: both; see
# tools/experiments/README.md) against the plain forward (f:) and the real one (e:) of the working tree
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2; do
  for shape in rbf:1024:64:64:4:2 lin:512:128:128:8:1 rbf:512:128:128:4:1; do
    echo -n "plain   "; python tools/ab.py --one new f:$shape 2>&1 | grep median
    for which in new expA expB expAB; do echo -n "edges   "; python tools/ab.py --one $which e:$shape 2>&1 | grep median; done
  done
done
