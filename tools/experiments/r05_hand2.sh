#!/bin/bash
# the handed-down edge pointer with its rare path out of line, against the build before it
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SK_AB_BASE=r05pre
python tools/ab.py e:rbf:1024:64:64:4:2 e:lin:512:128:128:8:1 e:rbf:512:128:128:4:1 e:rbf:512:128:128:8:1 e:lin:1024:128:128:4:1 e:rbf:1024:100:100:3:0 c4 g:lin:512:128:128:8:1 2>&1 | grep median
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
