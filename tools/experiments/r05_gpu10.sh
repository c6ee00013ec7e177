cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
R=$PWD; rm -rf gpurun_out/r05_reach; mkdir -p gpurun_out/r05_reach
for part in 0 1 2 3 4; do
  (cd /tmp; timeout 700 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r05_reach/p$part -o t -- python $R/tools/reach_sweep.py $part 5 2>&1 | grep -v amdgpu.ids | tail -2)
  find gpurun_out/r05_reach -name "*kernel_trace.csv" -delete
done
find gpurun_out/r05_reach -name "*kernel_stats.csv" | head
