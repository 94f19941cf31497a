#!/bin/bash
# kernel-trace of tools/kbench.py on the GPU box; prints per-kernel average durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_kb; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/kbench.py ${1:-C3} > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print("%-70s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"])/1e3))
PY
tail -1 $O/log.txt
