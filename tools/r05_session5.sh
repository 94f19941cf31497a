#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s5; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_fuzz.py tests/test_gpu_whole_matrix.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt
AB="build/ab/libprosstt_amd"
{
KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py C3 8 ${AB}_r4.so shipped ${AB}_k3h_grid512.so ${AB}_k3h_grid1024.so ${AB}_k3h_none.so
KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py T32 8 ${AB}_r4.so shipped ${AB}_k3h_grid512.so ${AB}_k3h_grid1024.so ${AB}_k3h_none.so
KBENCH_BURST=10 timeout 600 python3 tools/kbench_ab.py C4 6 ${AB}_r4.so shipped
KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py C2 8 ${AB}_r4.so shipped
} 2>&1 | grep -v amdgpu | tee $O/kbench.txt
