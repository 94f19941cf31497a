#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_whole_matrix.py -x -q 2>&1 | tail -2
timeout 900 python3 tools/kbench_ab.py C3 14 shipped build/ab/libprosstt_amd_plainstore.so build/ab/libprosstt_amd_philox7.so build/ab/libprosstt_amd_nostore.so build/ab/libprosstt_amd_k3h_noredo.so build/ab/libprosstt_amd_k3h_noheavy.so 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
cd /tmp && export TMPDIR=/tmp; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/kbench.py C3 > $O/trace.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("   %-60s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3))
PY
