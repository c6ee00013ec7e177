#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03n; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "rbf or symmetric or api or rescue or exploding or hip_graph or never_sync" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log | cut -c1-250
SK_AB_BASE=r03a python tools/r03_ab.py c4
