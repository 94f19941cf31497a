cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_build; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/build_configs.py C5 > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.3:
            print("%-70s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
tail -2 $O/log.txt
