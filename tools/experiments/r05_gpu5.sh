cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|^$" | tail -40
