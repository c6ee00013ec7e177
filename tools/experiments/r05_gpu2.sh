# A/B of the age-rank shares of mid-size no-queue launches (SK_FUSED_MID=0: equal shares), same box, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for mid in 0 1; do
    for cfg in "shard 64" "shard 96" "shard 128" "mmd 128" "mmd 256"; do
      echo -n "MID=$mid  "; SK_FUSED_MID=$mid python tools/experiments/r05_small_steps.py $cfg 100 2>&1 | grep -v amdgpu.ids
    done
  done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
bash tools/experiments/r05_timelines.sh r05_timelines_b 2>&1 | grep -v "amdgpu.ids" | head -150
