#!/usr/bin/env python3
"""(GPU box, CPU only) Why bench.py's oracle phases took 12 s in one run and 340 s in the next: time the same oracle call
repeatedly with the OpenMP threads pinned (OMP_PROC_BIND=spread, OMP_PLACES=threads: what bench.py set) and not pinned."""
import os, subprocess, sys, time
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    X = np.cumsum(rng.normal(size=(16, 128, 8)), 1) / 32
    Y = np.cumsum(rng.normal(size=(512, 128, 8)), 1) / 32
    nt = int(sys.argv[1])
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); O.gram_pipeline(X, Y, 0, 1.0, 1, nthreads=nt); ts.append(time.perf_counter() - t0)
    print("threads %3d bind %-6s places %-8s wait %-8s: %s" % (nt, os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), os.environ.get("OMP_WAIT_POLICY"),
                                                           " ".join("%.2f" % t for t in ts)), flush=True)
    sys.exit(0)
print(open("/sys/fs/cgroup/cpu.max").read().strip(), "| cpus visible", os.cpu_count(), "| affinity", len(os.sched_getaffinity(0)), "| loadavg", open("/proc/loadavg").read().strip())
for rnd in range(2):
    for nt in (16, 256):
        for env in ({"OMP_PROC_BIND": "spread", "OMP_PLACES": "threads"}, {"OMP_PROC_BIND": "false"}, {"OMP_PROC_BIND": "false", "OMP_WAIT_POLICY": "passive"}):
            e = {k: v for k, v in os.environ.items() if not k.startswith("OMP_")}
            e.update(env)
            subprocess.run([sys.executable, __file__, str(nt)], env=e)
print("loadavg", open("/proc/loadavg").read().strip())
