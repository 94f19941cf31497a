#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O; cd $R
timeout 900 python3 tools/kbench_ab.py C3 12 shipped build/ab/libprosstt_amd_prev.so build/ab/libprosstt_amd_philox7.so build/ab/libprosstt_amd_run40.so build/ab/libprosstt_amd_nostore.so build/ab/libprosstt_amd_s1.so build/ab/libprosstt_amd_s12.so build/ab/libprosstt_amd_s1_nostore.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
