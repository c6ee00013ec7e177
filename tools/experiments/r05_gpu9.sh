cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python tools/experiments/r05_thresholds.py second 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_thresholds2.txt
