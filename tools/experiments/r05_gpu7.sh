cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
SK_AB_BASE=r04 python tools/ab.py c5 c3 c2 mmd32 mmd64 mmd128 shard64 shard128 c4 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_r04_vs_r05.txt
cat gpurun_out/r05_ab_r04_vs_r05.txt
timeout 600 python tools/crossovers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_crossovers.txt; cat gpurun_out/r05_crossovers.txt
rm -rf gpurun_out/r05_reach; (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/gpurun_out/r05_reach -o t -- python $OLDPWD/tools/reach_sweep.py 2>&1 | grep -v amdgpu.ids | tail -3)
find gpurun_out/r05_reach -name "*kernel_trace.csv" -delete
