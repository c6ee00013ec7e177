cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_configs.py -x -q -m gpu -k "bands_of_a_pair or every_inline" 2>&1 | tail -2
timeout 600 python tools/experiments/r05_split.py quick 2>&1 | grep -v amdgpu.ids
