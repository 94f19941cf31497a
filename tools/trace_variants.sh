#!/bin/bash
# per-kernel average durations (rocprofv3 kernel trace) of tools/kbench.py for the shipped library and every build/ab variant
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for lib in "" build/ab/libprosstt_amd_*.so; do
  O=$R/gpurun_out/trace_var; rm -rf $O; mkdir -p $O
  if [ -n "$lib" ]; then export PROSSTT_AMD_LIB=$R/$lib; else unset PROSSTT_AMD_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/kbench.py ${1:-C3} > $O/log.txt 2>&1
  echo "== ${lib:-shipped}"
  python3 - <<PY
import csv, glob
for f in glob.glob("$O/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sample_counts" in r["Name"]:
            print("   %-60s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
