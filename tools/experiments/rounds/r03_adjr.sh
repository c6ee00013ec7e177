#!/bin/bash
# fused RBF adjoint (round 3): parity tests of the rewritten kernel + timing of a C4 step
set -u
OUT=$PWD/gpurun_out/r03b; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "rbf or symmetric or api or poisoned or long_first" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
timeout 600 python bench.py --config c4 --no-extras > $OUT/bench_c4.json 2> $OUT/bench_c4.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_c4 -o trace -- python $REPO/bench.py --config c4 --steps 2 --warmup 1 --no-extras > $OUT/trace_c4.json 2> $OUT/trace_c4.err
cd $REPO
python tools/r02_kstat.py $OUT/trace_c4 k_ > $OUT/c4_kernels.txt 2>&1
tail -15 $OUT/pytest.log; cat $OUT/c4_kernels.txt; cat $OUT/bench_c4.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('c4 ms/step', d['ms_per_step'])"; tail -3 $OUT/bench_c4.err
