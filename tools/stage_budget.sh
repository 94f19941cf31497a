#!/bin/bash
# Instruction counts of the streaming kernel by stage: one SQ counter pass of tools/kbench.py for the shipped library
# and for the timing-only builds that stop after stage 1 (s1) and after stage 2 (s12) of tools/ablate.py; the
# differences are what stages 2 and 3 issue.  K3h's counts come with the shipped pass.
# usage (on the GPU box, after `python3 tools/ablate.py s1 s12`): tools/stage_budget.sh [tag] [C3|T32|C4] -> gpurun_out/stage_budget_<tag>.txt
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r06}; CFG=${2:-C3}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stage_budget_$CFG; rm -rf $O; mkdir -p $O; cd $R
for v in shipped s1 s12; do
  if [ $v = shipped ]; then unset PROSSTT_AMD_LIB; else export PROSSTT_AMD_LIB=$R/build/ab/libprosstt_amd_$v.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES \
      --output-format csv -d $O/$v -- python3 tools/kbench.py $CFG > $O/$v.log 2>&1
done
unset PROSSTT_AMD_LIB
python3 - > $R/gpurun_out/stage_budget_$TAG.txt <<PY
import csv, glob, collections
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
tab = {}
for v in ("shipped", "s1", "s12"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % v, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "sample_counts" in k:
                agg["stream" if "stream" in k else "K3h"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            tab[(v, k)] = {n: sum(c[n]) / max(len(c[n]), 1) for n in names}
import sys
sys.path.insert(0, "$R")
from prosstt_amd import workloads
passes = workloads.CONFIGS["$CFG"]["N"] * workloads.CONFIGS["$CFG"]["G"] / 256.0      # passes of a wave over one cell x 256 genes
print("# wave-level instructions per launch at $CFG, SQ counters, tools/stage_budget.sh; per pass = / %.4g" % passes)
print("%-34s %10s %10s %10s %10s %10s %10s %10s | %10s %8s" % tuple(["kernel / build"] + [n[9:] for n in names] + ["sum", "per pass"]))
def row(label, d):
    tot = sum(d[n] for n in names)
    print("%-34s %10.4g %10.4g %10.4g %10.4g %10.4g %10.4g %10.4g | %10.4g %8.1f" % tuple([label] + [d[n] for n in names] + [tot, tot / passes]))
s, a, b = tab[("shipped", "stream")], tab[("s1", "stream")], tab[("s12", "stream")]
row("stream kernel, shipped", s)
row("  stage 1 alone (s1 build)", a)
row("  stages 1 + 2 (s12 build)", b)
row("  => stage 2 (s12 - s1)", {n: b[n] - a[n] for n in names})
row("  => stage 3 + drain (shipped - s12)", {n: s[n] - b[n] for n in names})
row("K3h, shipped", tab[("shipped", "K3h")])
PY
cat $R/gpurun_out/stage_budget_$TAG.txt
