#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SK_AB_BASE=r05pre
python tools/ab.py mmd32 mmd64 mmd128 g:lin:64:64:64:4:1 2>&1 | grep median
bash tools/experiments/r05_timelines.sh r05_timelines_fin > /dev/null 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
