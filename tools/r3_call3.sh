#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_whole_matrix.py tests/test_gpu_two_ranks.py -x -q -s > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
PROSSTT_BENCH_BACKEND=gloo PROSSTT_BENCH_ONE_GPU=1 timeout 1200 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench2_gloo.json 2> $O/bench2_gloo.err; tail -c 3000 $O/bench2_gloo.json; grep -v amdgpu.ids $O/bench2_gloo.err | tail -5
timeout 600 tools/microbench6 1024 > $O/microbench6_4waves.log 2>&1; tail -3 $O/microbench6_4waves.log
