#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sA; mkdir -p $O; cd $R
AB="build/ab/libprosstt_amd"
{
KBENCH_SORT=1 KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py C3 10 shipped ${AB}_base.so ${AB}_salu8.so ${AB}_valu4.so
} 2>&1 | grep -v amdgpu | tee $O/kbench.txt
