cd ${GRAFT_REPO_ROOT:-/root/repo}
SK_AB_BASE=r04 python tools/ab.py c5 c3 shard128 c4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ab2.txt
timeout 900 python -m pytest tests/test_configs.py -x -q -m gpu -k "loss_launch or merged_loss or bands_of_a_pair" 2>&1 | tail -3
