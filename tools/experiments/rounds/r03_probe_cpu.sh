nproc; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
lscpu | grep -E "Socket|Core|Thread|Model name|NUMA" 
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
