#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_whole_matrix.py tests/test_gpu_full_size.py tests/test_gpu_pipeline.py tests/test_gpu_sharding.py -x -q 2>&1 | tail -3
timeout 900 python3 tools/kbench_ab.py C3 14 shipped build/ab/libprosstt_amd_prev.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
