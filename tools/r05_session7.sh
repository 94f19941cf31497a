#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s7; mkdir -p $O; cd $R
PROSSTT_AMD_LIB=$R/build/ab/libprosstt_amd_k3h_trace.so KBENCH_ITERS=3 timeout 300 python3 tools/kbench.py C3 2>&1 | grep K3HTRACE | tail -16 | tee $O/trace_C3.txt
