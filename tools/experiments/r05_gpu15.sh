cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | tail -3
for seed in 601 602; do timeout 900 python tools/fuzz_api.py 150 $seed 2>&1 | grep -v amdgpu.ids | tail -1; done
