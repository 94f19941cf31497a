#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + PMC passes of the default bench command.
# usage: tools/profile_bench.sh <tag>     (summaries land in gpurun_out/prof_<tag>/)
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
ARGS="bench.py --steps 10 --warmup 2 --cpu-cells 0"
python3 $ARGS > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ARGS > $O/bench_trace.json 2> $O/trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_sq -- python3 $ARGS > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_fetch -- python3 $ARGS > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE SQ_WAVES --output-format csv -d $O/pmc_write -- python3 $ARGS > /dev/null 2> $O/pmc_write.err
python3 tools/summarize_prof.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
