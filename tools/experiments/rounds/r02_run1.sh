#!/bin/bash
# Round-2 GPU pass 1 (run through gpurun): full GPU suite, bench on c3 / c5 / c4, kernel traces of c5 and c4.
set -u
OUT=$PWD/gpurun_out/r02a; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -30 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 600 $OUT/bench_c3.err
timeout 900 python bench.py --config c5 --steps 5 --warmup 2 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 600 $OUT/bench_c5.err
timeout 900 python bench.py --config c4 --steps 3 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; tail -c 600 $OUT/bench_c4.err
timeout 300 python bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
cd /tmp
for cfg in c5 c4; do
  timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$cfg -o trace -- python $REPO/bench.py --config $cfg --steps 3 --warmup 1 --no-extras > $OUT/trace_$cfg.json 2> $OUT/trace_$cfg.err
done
cd $REPO
for cfg in c5 c4; do
  f=$(find $OUT/trace_$cfg -name "*kernel_stats.csv" | head -1)
  echo "== $cfg kernel stats =="; [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-90s %6s %14s %12s %6s"%(r["Name"][:90],r["Calls"],r["TotalDurationNs"],r["AverageNs"],r["Percentage"]))
PY
done > $OUT/trace_summary.txt 2>&1
cat $OUT/trace_summary.txt
for c in c3 c5 c4 c2; do echo "--- $c"; head -c 3000 $OUT/bench_$c.json; echo; done
