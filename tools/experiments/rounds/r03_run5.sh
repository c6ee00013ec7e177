#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03g; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log | cut -c1-300
for q in 100 40 25 10 1; do
    SK_FUSED_Q_STATIC=$q timeout 300 python bench.py --config c3 --steps 30 --warmup 10 --no-extras > $OUT/bench_c3_q$q.json 2> $OUT/bench_c3_q$q.err
    python -c "
import json; d=json.loads(open('$OUT/bench_c3_q$q.json').read().strip().split('\n')[-1]); print('c3 q=$q ms/step %.3f' % d['ms_per_step'])"
done
for q in 40 20 1; do
  SK_FUSED_Q_STATIC=$q timeout 300 python bench.py --config c4 --no-extras > $OUT/bench_c4_q$q.json 2> $OUT/bench_c4_q$q.err
  python -c "
import json; d=json.loads(open('$OUT/bench_c4_q$q.json').read().strip().split('\n')[-1]); print('c4 q=$q ms/step %.3f' % d['ms_per_step'])"
done
