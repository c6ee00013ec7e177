cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_configs.py tests/test_routes.py tests/test_gpu_parity.py -x -q -m gpu -k "rbf_adjoint or sym or routes or triang or second_argument" 2>&1 | tail -3
timeout 600 python tools/experiments/r05_yside_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_yside_ab_after.txt
