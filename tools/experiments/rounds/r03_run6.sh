#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03i; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_configs.py -m gpu -x -q -k "multiband or c5 or long" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log | cut -c1-250
python tools/r03_ab.py c5
for q in 100 60 35 15 1; do
  SK_FUSEDMB_Q_STATIC=$q python tools/r03_ab.py --one new c5
done
