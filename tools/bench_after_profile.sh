#!/bin/bash
# Runs on the GPU box AFTER tools/collect_profile.sh <tag> has put the round's summaries into profiles/ (and they have been
# committed): the bench lines then carry roofline.traffic from them (bench.py reads profiles/<tag>_summary.txt and
# profiles/<tag>_t32_summary.txt and compares their kernel_source_sha).  usage: tools/bench_after_profile.sh <tag> -> gpurun_out/after_<tag>/
TAG=${1:-r06}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/after_$TAG; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $R
ARGS="bench.py --steps 10 --warmup 2 --cpu-cells 0 --no-end-to-end --no-target-shape"
python3 $ARGS 2> $O/unprofiled.err | tail -1 > $O/bench_unprofiled.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ARGS 2> $O/trace.err | tail -1 > $O/bench_under_rocprof.json
python3 $ARGS --config T32 2> $O/unprofiled_t32.err | tail -1 > $O/t32_bench_unprofiled.json
python3 bench.py 2> $O/default.err | tail -1 > $O/bench_default.json
python3 bench.py 2> $O/default2.err | tail -1 > $O/bench_default_second_run.json
python3 -c "
import json
for f in ('bench_unprofiled', 'bench_under_rocprof', 't32_bench_unprofiled', 'bench_default', 'bench_default_second_run'):
    d = json.load(open('$O/%s.json' % f)); r = d['roofline']
    print(f, d['ms_per_step'], r['kernel_ms'], r['frac'], r['frac_whole_step'], r['traffic'], r['traffic_source'], (d.get('north_star_shape') or {}).get('frac'), (d.get('north_star_shape') or {}).get('traffic'))
"
rm -rf $O/trace
