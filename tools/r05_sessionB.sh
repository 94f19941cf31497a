#!/bin/bash
# the other configurations under the same profile script (kernel trace + eight PMC passes each)
R=$GRAFT_REPO_ROOT; cd $R
BENCH_EXTRA="--config C4" bash tools/profile_bench.sh r05_c4 > /dev/null 2>&1; grep "under the tracer\|derived" -A2 gpurun_out/prof_r05_c4/summary.txt | cut -c1-300
BENCH_EXTRA="--config C2" bash tools/profile_bench.sh r05_c2 > /dev/null 2>&1; grep "under the tracer\|derived" -A2 gpurun_out/prof_r05_c2/summary.txt | cut -c1-300
BENCH_EXTRA="--config C5 --cells-per-gpu 125000" bash tools/profile_bench.sh r05_c5share > /dev/null 2>&1; grep "under the tracer\|derived" -A2 gpurun_out/prof_r05_c5share/summary.txt | cut -c1-300
