#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_q; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/quickbench.py > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sample_counts" in r["Name"]:
            print("%-60s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3))
PY
grep "heavy frac" $O/log.txt
