#!/bin/bash
# the next-chunk precompute on the small launches and the training steps; fuzz and determinism on the build
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SK_AB_BASE=r05pre
python tools/ab.py mmd32 mmd64 mmd128 g:lin:512:128:128:8:1 g:rbf:512:128:128:4:1 g:rbf:1024:64:64:4:2 2>&1 | grep median
timeout 400 python tools/fuzz_api.py 400 4242 2>&1 | tail -1
timeout 300 python tools/det_sweep.py --families 2>&1 | tail -1
