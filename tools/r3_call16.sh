#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_whole_matrix.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -3
timeout 900 python3 tools/kbench_ab.py C3 16 shipped build/ab/libprosstt_amd_prev.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
timeout 600 python3 tools/kbench_ab.py C4 8 shipped build/ab/libprosstt_amd_prev.so 2>&1 | grep -v amdgpu | tee -a $O/ab.log
