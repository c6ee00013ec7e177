#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03e; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_configs.py -m gpu -x -q -k "never_synchronise or rescue or exploding or hip_graph or falls_back" > $OUT/pytest_new.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_new.log
tail -30 $OUT/pytest_new.log | cut -c1-300
