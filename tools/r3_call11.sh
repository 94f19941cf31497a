#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python3 tools/kbench_ab.py C3 14 shipped build/ab/libprosstt_amd_philox10.so build/ab/libprosstt_amd_k3h_none.so build/ab/libprosstt_amd_k3h_noredo.so build/ab/libprosstt_amd_k3h_noheavy.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
