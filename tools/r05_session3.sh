#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_fuzz.py tests/test_gpu_whole_matrix.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.txt
AB="build/ab/libprosstt_amd"
{
KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py C3 8 ${AB}_r4.so shipped
KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py T32 8 ${AB}_r4.so shipped
KBENCH_BURST=10 timeout 600 python3 tools/kbench_ab.py C4 6 ${AB}_r4.so shipped
KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py C2 8 ${AB}_r4.so shipped
} 2>&1 | grep -v amdgpu | tee $O/kbench.txt
cd /tmp && export TMPDIR=/tmp; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/kbench.py C3 > $O/trace.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-60s calls %4s avg %10.1f us" % (row["Name"].split("(")[0][-60:], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
rm -rf $O/trace
