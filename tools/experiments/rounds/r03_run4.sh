#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03f; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log | cut -c1-300
for q in 100 85 70 55 40; do
  for cfg in c3; do
    SK_FUSED_Q_STATIC=$q timeout 300 python bench.py --config $cfg --steps 30 --warmup 10 --no-extras > $OUT/bench_${cfg}_q$q.json 2> $OUT/bench_${cfg}_q$q.err
    python -c "
import json; d=json.loads(open('$OUT/bench_${cfg}_q$q.json').read().strip().split('\n')[-1]); print('$cfg q=$q ms/step %.3f' % d['ms_per_step'])"
  done
done
for q in 100 70; do
  SK_FUSED_Q_STATIC=$q timeout 300 python bench.py --config c4 --no-extras > $OUT/bench_c4_q$q.json 2> $OUT/bench_c4_q$q.err
  python -c "
import json; d=json.loads(open('$OUT/bench_c4_q$q.json').read().strip().split('\n')[-1]); print('c4 q=$q ms/step %.3f' % d['ms_per_step'])"
  SK_FUSED_Q_STATIC=$q timeout 300 python bench.py --config c2 --steps 50 --warmup 10 --no-extras > $OUT/bench_c2_q$q.json 2> $OUT/bench_c2_q$q.err
  python -c "
import json; d=json.loads(open('$OUT/bench_c2_q$q.json').read().strip().split('\n')[-1]); print('c2 q=$q ms/step %.4f' % d['ms_per_step'])"
done
