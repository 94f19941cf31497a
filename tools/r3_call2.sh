#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O; cd $R
timeout 600 tools/microbench6 1024 > $O/microbench6_4waves.log 2>&1
timeout 600 tools/microbench6 512 > $O/microbench6_2waves.log 2>&1
timeout 600 python3 tools/api_time.py > $O/api_time.log 2>&1; cat $O/api_time.log
timeout 900 python3 bench.py --steps 10 --warmup 2 > $O/bench1.json 2> $O/bench1.err; tail -c 1500 $O/bench1.json; tail -3 $O/bench1.err
PROSSTT_BENCH_BACKEND=gloo PROSSTT_BENCH_ONE_GPU=1 timeout 1200 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench2_gloo.json 2> $O/bench2_gloo.err; tail -c 2500 $O/bench2_gloo.json; tail -5 $O/bench2_gloo.err
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
