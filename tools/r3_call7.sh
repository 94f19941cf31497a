#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_full_size.py tests/test_gpu_whole_matrix.py -x -q 2>&1 | tail -3
timeout 900 python3 tools/kbench_ab.py C3 12 shipped build/ab/libprosstt_amd_base.so build/ab/libprosstt_amd_ntstore.so build/ab/libprosstt_amd_philox7.so build/ab/libprosstt_amd_run40.so build/ab/libprosstt_amd_nostore.so build/ab/libprosstt_amd_s1.so build/ab/libprosstt_amd_s12.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
