#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R
timeout 900 python3 tools/kbench_ab.py C3 14 shipped build/ab/libprosstt_amd_prev.so build/ab/libprosstt_amd_cload.so build/ab/libprosstt_amd_cload_storefirst.so build/ab/libprosstt_amd_plainstore.so build/ab/libprosstt_amd_cload_plainstore.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
