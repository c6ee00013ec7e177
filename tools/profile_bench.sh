#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + stats of the bench command, then separate
# PMC passes (counters are never combined with API tracing) restricted to the solver kernel.
# Usage: tools/profile_bench.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
ARGS=${*:-"--steps 5 --warmup 2"}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp

rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- python "$REPO/bench.py" $ARGS > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"

PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $set | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_fwd_fused|k_fwd_wave|k_fwd_simple|k_adj|k_increments|k_static" -f csv \
      -d "$OUT/pmc_$name" -o pmc -- python "$REPO/bench.py" $PMC_ARGS > /dev/null 2> "$OUT/pmc_$name.err" || echo "pmc set failed: $set" >> "$OUT/failed.txt"
done
cd "$REPO"
python tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
