#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s11; mkdir -p $O; cd $R
cd /tmp && export TMPDIR=/tmp; cd $R
for v in r4 shipped; do
  LIB=$([ $v = shipped ] && echo shipped || echo build/ab/libprosstt_amd_$v.so)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES \
      --output-format csv -d $O/$v -- python3 tools/kbench_ab.py C3 4 $LIB > $O/$v.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES \
      --output-format csv -d $O/${v}_b -- python3 tools/kbench_ab.py C3 4 $LIB > $O/${v}_b.log 2>&1
done
python3 - <<PY | tee $O/summary.txt
import csv, glob, collections
for v in ("r4", "r4_b", "shipped", "shipped_b"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % v, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "sample_counts" in k: agg[k[-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            for n, xs in sorted(c.items()): print("%-10s %-42s %-22s %.6g (n=%d)" % (v, k, n, sum(xs)/len(xs), len(xs)))
PY
rm -rf $O/r4 $O/r4_b $O/shipped $O/shipped_b
