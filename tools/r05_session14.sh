#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s14; mkdir -p $O; cd $R
PROSSTT_AMD_LIB=$R/build/ab/libprosstt_amd_cold_trace.so timeout 300 python3 tools/cold_probe.py 2>&1 | grep -v amdgpu | tee $O/cold_trace_first_process.txt
PROSSTT_AMD_LIB=$R/build/ab/libprosstt_amd_cold_trace.so timeout 300 python3 tools/cold_probe.py 2>&1 | grep -v amdgpu | grep -B3 -A12 "deferred check" | tee $O/cold_trace_second_process.txt
