#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O; cd $R
timeout 900 python3 tools/kbench_ab.py C3 10 shipped build/ab/libprosstt_amd_s1.so build/ab/libprosstt_amd_s12.so build/ab/libprosstt_amd_s1_nostore.so build/ab/libprosstt_amd_s1_nostore_nophilox.so build/ab/libprosstt_amd_s1_nopush.so build/ab/libprosstt_amd_nostore.so build/ab/libprosstt_amd_philox7.so build/ab/libprosstt_amd_base.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
timeout 1500 bash tools/pmc_kbench.sh C3 r3e > /dev/null 2>&1; grep stream gpurun_out/pmc_r3e/summary.txt
