#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
timeout 300 python3 tools/cold_probe.py 2>&1 | grep -v amdgpu | tee $O/cold_probe.txt
timeout 300 python3 tools/cold_probe.py 2>&1 | grep -v amdgpu | tee $O/cold_probe_second_process.txt
bash tools/pmc_kbench.sh T32 t32 > /dev/null 2>&1; cp gpurun_out/pmc_t32/summary.txt $O/pmc_T32.txt
cd /tmp && export TMPDIR=/tmp; cd $R
for cfg in T32 C3; do for srt in 0 1; do
  KBENCH_SORT=$srt rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/fetch_${cfg}_$srt -- python3 tools/kbench.py $cfg > $O/fetch_${cfg}_$srt.log 2>&1
done; done
rocprofv3 --pmc WRITE_SIZE SQ_WAVES --output-format csv -d $O/write_T32 -- python3 tools/kbench.py T32 > $O/write_T32.log 2>&1
python3 - > $O/fetch_write.txt <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/fetch_*") + glob.glob("$O/write_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "sample_counts" in k: agg[k[-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, xs in sorted(v.items()): print("%-28s %-42s %-18s %.6g  (n=%d)" % (d.split("/")[-1], k, c, sum(xs)/len(xs), len(xs)))
PY
cat $O/fetch_write.txt
for c in T32 C3; do timeout 600 python3 tools/workload_stats.py $c 2>&1 | grep -v amdgpu; done | tee $O/workload_stats.txt
rm -rf $O/fetch_*/ $O/write_*/ gpurun_out/pmc_t32/[abcd]
