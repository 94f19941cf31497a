#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + PMC passes of the default bench command.
# usage: [BENCH_EXTRA="--config T32"] tools/profile_bench.sh <tag>     (summaries land in gpurun_out/prof_<tag>/)
# The default set is taken WITHOUT the extra case on north_star's shape (--no-target-shape): the kernel stats then hold the
# headline workload's dispatches only; the T32 set is its own run (BENCH_EXTRA="--config T32").
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
ARGS="bench.py --steps 10 --warmup 2 --cpu-cells 0 --no-end-to-end --no-target-shape $BENCH_EXTRA"
PMC="$ARGS --ramp-ms 0"     # counter passes: no clock ramp (instruction and byte counts do not depend on the clock; 18 dispatches per kernel instead of 250)
python3 $ARGS > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ARGS > $O/bench_trace.json 2> $O/trace.err
# the wave-cycle budget of the SQ (8 counters per pass): WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_sq -- python3 $PMC > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq2 -- python3 $PMC > /dev/null 2> $O/pmc_sq2.err
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_sq3 -- python3 $PMC > /dev/null 2> $O/pmc_sq3.err
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_CYCLES --output-format csv -d $O/pmc_sq4 -- python3 $PMC > /dev/null 2> $O/pmc_sq4.err
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_fetch -- python3 $PMC > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE SQ_WAVES --output-format csv -d $O/pmc_write -- python3 $PMC > /dev/null 2> $O/pmc_write.err
python3 tools/summarize_prof.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
