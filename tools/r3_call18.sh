#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3r; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_whole_matrix.py tests/test_gpu_full_size.py -x -q -s 2>&1 | grep -v amdgpu | grep "list\]\|passed\|failed\|Error\|assert" | tail -6
timeout 900 python3 tools/kbench_ab.py C3 16 shipped build/ab/libprosstt_amd_prev.so build/ab/libprosstt_amd_bail3.so build/ab/libprosstt_amd_bail10.so build/ab/libprosstt_amd_bail16.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
