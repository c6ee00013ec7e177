#!/bin/bash
# resident waves per CU of the one-band fused forward on the small launches (SK_FUSED_WPC; the launcher's own choice is 8 at C2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2; do
  for wpc in 0 8 12 16; do
    for cfg in c2 mmd32 mmd64 mmd128 shard64; do
      echo -n "wpc $wpc  "; SK_FUSED_WPC=$wpc python tools/ab.py --one new $cfg 2>&1 | grep median
    done
  done
done
