#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s4; mkdir -p $O; cd $R
AB="build/ab/libprosstt_amd"
{
KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py C3 8 shipped ${AB}_k3h_grid512.so ${AB}_k3h_grid768.so ${AB}_k3h_grid1024.so ${AB}_k3h_grid2048.so ${AB}_k3h_none.so ${AB}_k3h_noheavy.so ${AB}_k3h_noredo.so ${AB}_r4.so
KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py T32 8 shipped ${AB}_k3h_grid512.so ${AB}_k3h_grid768.so ${AB}_k3h_grid1024.so ${AB}_k3h_grid2048.so ${AB}_k3h_none.so ${AB}_k3h_noheavy.so ${AB}_k3h_noredo.so ${AB}_r4.so
} 2>&1 | grep -v amdgpu | tee $O/kbench.txt
