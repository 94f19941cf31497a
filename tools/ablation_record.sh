#!/bin/bash
# Runs on the GPU box: the timing-only and tuning variants of tools/ablate.py against the shipped library, interleaved
# in one process (tools/kbench_ab.py).  usage: tools/ablation_record.sh <tag>  -> gpurun_out/ablation_<tag>.txt
TAG=${1:-r06}; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
  echo "# Variants of tools/ablate.py on C3, interleaved with the shipped library in ONE process (tools/kbench_ab.py: 10 rounds of 20 calls back to back,"
  echo "# kernel = HIP-event time of the stream kernel in a round's last call, call = prep + stream kernel + K3h, mean of the round's calls).  Outputs of the timing-only builds are"
  echo "# wrong by construction (see the sums); only the times are read.  kernel_source_sha: $(python3 -c 'import bench; print(bench.kernel_source_sha())')"
  KBENCH_BURST=20 timeout 1500 python3 tools/kbench_ab.py C3 10 shipped $(ls build/ab/libprosstt_amd_*.so) 2>&1 | grep -v amdgpu
} > gpurun_out/ablation_$TAG.txt
cat gpurun_out/ablation_$TAG.txt
