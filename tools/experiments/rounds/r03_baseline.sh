#!/bin/bash
# Round-3 start-of-round baseline on a fresh box: GPU suite, bench c3 / c4 / c2 / c5, kernel trace of c4.  Output: gpurun_out/r03a/
set -u
OUT=$PWD/gpurun_out/r03a; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
for cfg in c3 c4 c2 c5; do
  timeout 600 python bench.py --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_c4 -o trace -- python $REPO/bench.py --config c4 --steps 2 --warmup 1 --no-extras > $OUT/trace_c4.json 2> $OUT/trace_c4.err
cd $REPO
python tools/r02_kstat.py $OUT/trace_c4 k_ > $OUT/c4_kernels.txt 2>&1
tail -3 $OUT/pytest.log; cat $OUT/c4_kernels.txt; for cfg in c3 c4 c2 c5; do python - $OUT/bench_$cfg.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print(d["config"]["name"], "ms/step %.3f"%d["ms_per_step"], "value %.4g"%d["value"], "roofline frac", d.get("roofline",{}).get("frac"))
except Exception as e: print("bench parse failed", sys.argv[1], e)
PY
done
