#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03d; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
timeout 900 python tools/r03_shard_times.py $OUT/c4_shard_times.json > $OUT/shard.log 2>&1
for cfg in c3 c4; do timeout 600 python bench.py --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; done
tail -4 $OUT/pytest.log; tail -6 $OUT/shard.log | cut -c1-400
for cfg in c3 c4; do python - $OUT/bench_$cfg.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print(d["config"]["name"], "ms/step %.3f"%d["ms_per_step"], "roofline frac", d.get("roofline",{}).get("frac"), "cpu", {k:d.get("cpu_baseline",{}).get(k) for k in ("value","single_thread_value","speedup_over_1_thread","threads_used")})
except Exception as e: print("bench parse failed", sys.argv[1], e)
PY
done
