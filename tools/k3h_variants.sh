#!/bin/bash
# GPU box: K3h's own duration (rocprofv3 kernel trace of tools/kbench.py) for the shipped library and build/ab variants.
# usage: tools/k3h_variants.sh <tag> "<cfgs>" variant...
TAG=$1; CFGS=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
for cfg in $CFGS; do
  for v in shipped "$@"; do
    if [ $v = shipped ]; then unset PROSSTT_AMD_LIB; else export PROSSTT_AMD_LIB=$R/build/ab/libprosstt_amd_$v.so; fi
    KBENCH_SORT=1 KBENCH_ITERS=${KBENCH_ITERS:-40} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_${cfg}_$v -- python3 tools/kbench.py $cfg > $O/kb_${cfg}_$v.log 2>&1
    f="$(ls -t $O/t_${cfg}_$v/*/*kernel_stats.csv | head -1)"
    python3 - "$f" $cfg $v <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0].split("::")[-1][:28]: r for r in csv.DictReader(open(sys.argv[1])) if "k3::" in r["Name"]}
print("%-4s %-16s " % (sys.argv[2], sys.argv[3]) + " | ".join("%s avg %.1f min %.1f us" % (k, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3) for k, r in sorted(rows.items())))
PY
    grep K3HTRACE $O/kb_${cfg}_$v.log | head -12
    rm -rf $O/t_${cfg}_$v
  done
done
