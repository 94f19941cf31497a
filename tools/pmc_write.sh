#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_w; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/a -- python3 tools/kbench.py ${1:-C3} > $O/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/a/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-34:]
        if "sample_counts" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, xs in sorted(v.items()): print("%-36s %-24s %.5g" % (k, c, sum(xs)/len(xs)))
PY
