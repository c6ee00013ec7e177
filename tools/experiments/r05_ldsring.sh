#!/bin/bash
# the lanes' look-up of a drawn chunk's first pair through an LDS copy of the chunk ring, against the build before it
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SK_AB_BASE=r05pre
python tools/ab.py e:rbf:1024:64:64:4:2 e:lin:512:128:128:8:1 e:rbf:512:128:128:4:1 c4 c3 2>&1 | grep median
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
