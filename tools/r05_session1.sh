#!/bin/bash
# Round 5, first GPU session: where does the 32-branch tree lose its 15 % per sample?  (T32 = C4's tree, 50 000 cells.)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s1; mkdir -p $O; cd $R
AB="build/ab/libprosstt_amd"
timeout 300 python3 tools/cold_start.py C3 2>&1 | grep -v amdgpu > $O/cold_C3.txt; cat $O/cold_C3.txt
{
KBENCH_BURST=20 timeout 900 python3 tools/kbench_ab.py T32 8 shipped ${AB}_noload.so ${AB}_nostore.so ${AB}_xcd_transpose.so ${AB}_order_stripmajor.so ${AB}_load_aux2.so ${AB}_load_aux16.so ${AB}_s1.so ${AB}_s12.so
echo "# cells in the order of their mean-tensor rows (KBENCH_SORT=1):"
KBENCH_SORT=1 KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py T32 8 shipped ${AB}_noload.so ${AB}_xcd_transpose.so
KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py C3 8 shipped ${AB}_noload.so
KBENCH_SORT=1 KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py C3 8 shipped ${AB}_noload.so
KBENCH_BURST=10 timeout 600 python3 tools/kbench_ab.py C4 6 shipped ${AB}_xcd_transpose.so
KBENCH_SORT=1 KBENCH_BURST=10 timeout 600 python3 tools/kbench_ab.py C4 6 shipped ${AB}_xcd_transpose.so
} 2>&1 | grep -v amdgpu | tee $O/kbench.txt
bash tools/pmc_kbench.sh T32 t32 > /dev/null 2>&1; cp gpurun_out/pmc_t32/summary.txt $O/pmc_T32.txt
cd /tmp && export TMPDIR=/tmp; cd $R
for cfg in T32 C3; do for srt in 0 1; do
  KBENCH_SORT=$srt rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/fetch_${cfg}_$srt -- python3 tools/kbench.py $cfg > $O/fetch_${cfg}_$srt.log 2>&1
done; done
rocprofv3 --pmc WRITE_SIZE SQ_WAVES --output-format csv -d $O/write_T32 -- python3 tools/kbench.py T32 > $O/write_T32.log 2>&1
python3 - > $O/fetch_write.txt <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/fetch_*") + glob.glob("$O/write_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-34:]
            if "sample_counts" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, xs in sorted(v.items()): print("%-28s %-36s %-18s %.6g  (n=%d)" % (d.split("/")[-1], k, c, sum(xs)/len(xs), len(xs)))
PY
cat $O/fetch_write.txt
bash tools/stage_budget.sh r05_t32 T32 > /dev/null 2>&1; cat gpurun_out/stage_budget_r05_t32.txt | cut -c1-170
bash tools/stage_budget.sh r05_c3 C3 > /dev/null 2>&1; cat gpurun_out/stage_budget_r05_c3.txt | cut -c1-170
for c in T32 C3; do timeout 900 python3 tools/list_stats.py $c 2>&1 | grep -v amdgpu | tail -3; done | tee $O/list_stats.txt
rm -rf $O/fetch_*/ $O/write_*/ gpurun_out/pmc_t32/[abcd] gpurun_out/stage_budget_*/
