mkdir -p gpurun_out/r04h
python -m pytest tests/test_gpu_parity.py tests/test_configs.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r04h/pytest.txt
for r in 1 2 3; do
  for h in 0 1; do
    SK_FUSED_HALFW=$h python bench.py --config c3 --steps 30 --warmup 10 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('halfw=$h c3 ms', d['ms_per_step'])"
  done
done > gpurun_out/r04h/ab.txt 2>&1
for h in 0 1; do SK_FUSED_HALFW=$h python tools/ab.py f:linear:1024:64:64:8:2 2>/dev/null | grep new | tail -1 | sed "s/^/halfw=$h /"; done >> gpurun_out/r04h/ab.txt 2>&1
tail -3 gpurun_out/r04h/pytest.txt; cat gpurun_out/r04h/ab.txt
