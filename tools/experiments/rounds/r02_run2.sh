#!/bin/bash
# Round-2 GPU pass 2: full GPU suite on the current build, the N>1 code path on one GPU (RCCL, --force-dist), C5 bench.
set -u
OUT=$PWD/gpurun_out/r02b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c3_dist.json 2> $OUT/bench_c3_dist.err; tail -c 400 $OUT/bench_c3_dist.err
timeout 600 python bench.py --force-dist --config c4 --steps 2 --warmup 1 --no-extras > $OUT/bench_c4_dist.json 2> $OUT/bench_c4_dist.err; tail -c 400 $OUT/bench_c4_dist.err
timeout 600 python bench.py --config c5 --steps 5 --warmup 2 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 400 $OUT/bench_c5.err
for f in bench_c3_dist bench_c4_dist bench_c5; do echo "--- $f"; head -c 1500 $OUT/$f.json; echo; done
