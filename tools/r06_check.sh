#!/bin/bash
# GPU box: parity tests of the sampler, then per-kernel times (rocprofv3 kernel trace) of kbench on C3 and T32.
# usage: tools/r06_check.sh <tag> [pytest args]
TAG=${1:-s1}; shift; KB_CFGS=${KB_CFGS-C3 T32}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
if [ $# -eq 0 ]; then set -- tests/test_gpu_hw_math.py tests/test_gpu_sampler.py tests/test_gpu_fuzz.py; fi
timeout 1500 python3 -m pytest "$@" -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
for cfg in $KB_CFGS; do
  KBENCH_SORT=1 KBENCH_ITERS=40 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -- python3 tools/kbench.py $cfg > $O/kbench_$cfg.log 2>&1
  grep "call median" $O/kbench_$cfg.log
  f="$(ls -t $O/trace_$cfg/*/*kernel_stats.csv | head -1)"; cp "$f" $O/kernel_stats_$cfg.csv; grep -E "k3::|prep_kernel" "$f" | sed -E 's/\(.*\)"//' | cut -c1-120
done
