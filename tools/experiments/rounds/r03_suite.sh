#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03s; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
