#!/bin/bash
# PRNB-3 kernel: correctness first, then time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_sampler.py -x -q > $O/pytest_sampler.log 2>&1; tail -15 $O/pytest_sampler.log
timeout 300 python3 tools/kbench.py C3 > $O/kbench.log 2>&1; grep -v amdgpu $O/kbench.log | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; tail -15 $O/pytest_all.log
