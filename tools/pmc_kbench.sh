#!/bin/bash
# SQ counter passes of tools/kbench.py (the count sampler alone) -- the wave-cycle budget of the stream kernel.
# usage (on the GPU box): tools/pmc_kbench.sh [C3] [tag]   -> gpurun_out/pmc_kb_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp
CFG=${1:-C3}; TAG=${2:-kb}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 tools/kbench.py $CFG > $O/log.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -- python3 tools/kbench.py $CFG >> $O/log.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/c -- python3 tools/kbench.py $CFG >> $O/log.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_CYCLES --output-format csv -d $O/d -- python3 tools/kbench.py $CFG >> $O/log.txt 2>&1
python3 - > $O/summary.txt <<PY
import csv, glob, collections
for d in ("a","b","c","d"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "sample_counts" in k: agg[k[-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, xs in sorted(v.items()): print("%-42s %-24s %.6g  (n=%d)" % (k, c, sum(xs)/len(xs), len(xs)))
PY
cat $O/summary.txt
