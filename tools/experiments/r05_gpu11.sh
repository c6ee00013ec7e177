cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|^$" | tail -12
timeout 300 python tools/experiments/r05_yside_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_yside_ab.txt
timeout 600 python tools/crossovers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_crossovers.txt; cat gpurun_out/r05_crossovers.txt
timeout 900 python bench.py --gpus 1 --force-dist --steps 10 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/r05_bench_force_dist.json 2> gpurun_out/r05_bench_force_dist.err; tail -c 1500 gpurun_out/r05_bench_force_dist.json
