cd ${GRAFT_REPO_ROOT:-/root/repo}
for dr in 24 12 6 3; do echo "SK_FUSED_EDGE_DRAWS=$dr"; SK_FUSED_EDGE_DRAWS=$dr python tools/ab.py --one new c4 2>&1 | grep -v amdgpu.ids; SK_FUSED_EDGE_DRAWS=$dr python tools/ab.py --one new mmd128 2>&1 | grep -v amdgpu.ids; done
