#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s15; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_sampler.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python3 tools/cold_probe.py 2>&1 | grep -v amdgpu | tee $O/cold_probe.txt
cd /tmp && export TMPDIR=/tmp; cd $R
for c in C2 C3; do
KBENCH_BURST=20 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$c -- python3 tools/kbench_ab.py $c 4 shipped > $O/trace_$c.log 2>&1
python3 tools/gap_trace.py $O/trace_$c | tee $O/gaps_$c.txt
done
rm -rf $O/trace_C2 $O/trace_C3
