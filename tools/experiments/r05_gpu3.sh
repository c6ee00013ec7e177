cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_configs.py -x -q -m gpu -k "loss_launch or merged_loss" 2>&1 | tail -3
for i in 1 2; do
  for cfg in "mmd 32" "mmd 64"; do
    echo -n "default      "; python tools/experiments/r05_small_steps.py $cfg 200 2>&1 | grep -v amdgpu.ids
    echo -n "FUSED_WPC=12 "; SK_FUSED_WPC=12 python tools/experiments/r05_small_steps.py $cfg 200 2>&1 | grep -v amdgpu.ids
    echo -n "FUSED_WPC=4  "; SK_FUSED_WPC=4 python tools/experiments/r05_small_steps.py $cfg 200 2>&1 | grep -v amdgpu.ids
    echo -n "no launch    "; SK_NO_LOSS_LAUNCH=1 python tools/experiments/r05_small_steps.py $cfg 200 2>&1 | grep -v amdgpu.ids
  done
done
bash tools/experiments/r05_timelines.sh r05_timelines_c 2>&1 | grep -v "amdgpu.ids" | head -60
