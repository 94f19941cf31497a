#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O; cd $R
timeout 900 python3 tools/kbench_ab.py C3 16 shipped build/ab/libprosstt_amd_k3h_grid1024.so build/ab/libprosstt_amd_k3h_grid1536.so build/ab/libprosstt_amd_k3h_grid3072.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
