cd ${GRAFT_REPO_ROOT:-/root/repo}
for lead in 2 4 8; do echo "== SK_FUSEDMB_LEAD=$lead"; SK_FUSEDMB_LEAD=$lead timeout 600 python tools/experiments/r05_split.py quick 2>&1 | grep -v amdgpu.ids; done
timeout 600 python tools/experiments/r05_split.py 2>&1 | grep -v amdgpu.ids
