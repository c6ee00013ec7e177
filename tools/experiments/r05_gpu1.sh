set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_configs.py -x -q -m gpu -k "loss_launch or merged_loss" 2>&1 | tail -15
bash tools/experiments/r05_timelines.sh r05_timelines_a 2>&1 | grep -v "amdgpu.ids" | head -150
