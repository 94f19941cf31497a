#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s17; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_fuzz.py tests/test_gpu_whole_matrix.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
AB="build/ab/libprosstt_amd"
{
for c in C3 T32; do
KBENCH_SORT=1 KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py $c 8 ${AB}_r4.so shipped
done
KBENCH_SORT=1 KBENCH_BURST=10 timeout 600 python3 tools/kbench_ab.py C4 6 ${AB}_r4.so shipped
KBENCH_SORT=1 KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py C2 8 ${AB}_r4.so shipped
} 2>&1 | grep -v amdgpu | tee $O/kbench.txt
