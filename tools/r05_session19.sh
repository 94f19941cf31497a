#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s19; mkdir -p $O; cd $R
AB="build/ab/libprosstt_amd"
{
for c in C3 T32; do
KBENCH_SORT=1 KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py $c 10 shipped ${AB}_run40b.so ${AB}_run44.so ${AB}_run48b.so ${AB}_run48_bail10.so ${AB}_run40_bail10.so
done
KBENCH_SORT=1 KBENCH_BURST=10 timeout 600 python3 tools/kbench_ab.py C4 6 shipped ${AB}_run40b.so ${AB}_run48b.so ${AB}_run40_bail10.so
} 2>&1 | grep -v amdgpu | tee $O/kbench.txt
