#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_kb; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $O/a -- python3 tools/kbench.py ${1:-C3} > $O/log.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -- python3 tools/kbench.py ${1:-C3} >> $O/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a","b"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-34:]
            if "sample_counts" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, xs in sorted(v.items()): print("%-36s %-24s %.5g" % (k, c, sum(xs)/len(xs)))
PY
