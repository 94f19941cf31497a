#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu | tail -12
PROSSTT_BENCH_BACKEND=gloo PROSSTT_BENCH_ONE_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench2_gloo.json 2> $O/bench2_gloo.err; tail -c 2500 $O/bench2_gloo.json; grep -v "amdgpu.ids\|hostname" $O/bench2_gloo.err | tail -5
