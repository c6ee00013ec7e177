cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -2
SK_AB_BASE=r04 python tools/ab.py c4 c5 mmd64 c3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ab5.txt | awk '{print $1,$2,$4}' | paste - - - - - -
