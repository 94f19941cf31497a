#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s18; mkdir -p $O; cd $R
AB="build/ab/libprosstt_amd"
{
for c in C3 T32; do
KBENCH_SORT=1 KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py $c 8 shipped ${AB}_skipload.so ${AB}_run16b.so ${AB}_run24.so ${AB}_run40b.so ${AB}_bail10b.so ${AB}_bail3b.so
done
} 2>&1 | grep -v amdgpu | tee $O/kbench.txt
