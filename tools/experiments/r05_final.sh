#!/bin/bash
# the round's final evidence pass on ONE box: rocprofv3 summaries + PMC + shard times + bench lines (tools/profile_round.sh), step
# timelines, same-box A/B against the round-4 build, the crossover table, the long-first / short-second asymmetry grid
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_round.sh r05 > gpurun_out/r05_final_profile.log 2>&1
bash tools/experiments/r05_timelines.sh r05_timelines_final > /dev/null 2>&1
SK_AB_BASE=r04 python tools/ab.py c3 c2 c4 c5 mmd32 mmd64 mmd128 shard64 shard128 > gpurun_out/r05_ab_final.txt 2>&1
python tools/crossovers.py > gpurun_out/r05_crossovers.txt 2>&1
python tools/experiments/r05_asym.py > gpurun_out/r05_asym_final.txt 2>&1
