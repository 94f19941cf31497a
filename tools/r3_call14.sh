#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O; cd $R
timeout 900 python3 tools/kbench_ab.py C3 16 shipped build/ab/libprosstt_amd_prev.so build/ab/libprosstt_amd_pkonly.so build/ab/libprosstt_amd_phonly.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
