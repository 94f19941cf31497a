#!/bin/bash
# round 3, GPU call 1: baseline tests, instruction-cost table, workload dump, counter passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 300 tools/microbench5 1024 > $O/microbench5_4waves.log 2>&1
timeout 300 tools/microbench5 256 > $O/microbench5_1wave.log 2>&1
timeout 300 tools/microbench5 512 > $O/microbench5_2waves.log 2>&1
cat $O/microbench5_4waves.log
timeout 600 python3 tools/dump_workload.py C3 $O/c3_dump.npz 4096 > $O/dump.log 2>&1; tail -2 $O/dump.log
timeout 600 python3 tools/workload_stats.py > $O/workload_stats.log 2>&1; cat $O/workload_stats.log
timeout 1500 bash tools/pmc_kbench.sh C3 r3a
