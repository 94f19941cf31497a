#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_whole_matrix.py tests/test_gpu_full_size.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
timeout 900 python3 tools/kbench_ab.py C3 14 shipped build/ab/libprosstt_amd_prev.so build/ab/libprosstt_amd_strip32.so build/ab/libprosstt_amd_strip128.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
timeout 600 python3 tools/kbench_ab.py C4 8 shipped build/ab/libprosstt_amd_prev.so 2>&1 | grep -v amdgpu | tee -a $O/ab.log
KBENCH_CELLS=125000 timeout 600 python3 tools/kbench_ab.py C5 8 shipped build/ab/libprosstt_amd_prev.so 2>&1 | grep -v amdgpu | tee -a $O/ab.log
timeout 600 python3 tools/kbench_ab.py C2 8 shipped build/ab/libprosstt_amd_prev.so 2>&1 | grep -v amdgpu | tee -a $O/ab.log
