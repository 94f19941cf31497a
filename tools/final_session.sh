#!/bin/bash
# Runs on the GPU box: the round's closing measurements.  usage: tools/final_session.sh <tag>
TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_$TAG; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_bench.sh $TAG > /dev/null 2>&1; grep "under the tracer\|derived" -A2 gpurun_out/prof_$TAG/summary.txt | cut -c1-300
python3 bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
PROSSTT_BENCH_BACKEND=gloo PROSSTT_BENCH_ONE_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 5 --warmup 2 2> $O/bench2.err | tail -1 > $O/bench_2ranks_gloo_one_gpu.json
for c in C2 C4 C5; do KBENCH_CELLS=$([ $c = C5 ] && echo 125000 || echo 0) ; if [ $c = C5 ]; then export KBENCH_CELLS=125000; else unset KBENCH_CELLS; fi; timeout 600 python3 tools/kbench_ab.py $c 10 shipped 2>&1 | grep -v amdgpu; done | tee $O/other_configs.txt
bash tools/stage_budget.sh $TAG > /dev/null 2>&1; tail -7 gpurun_out/stage_budget_$TAG.txt | cut -c1-170
bash tools/ablation_record.sh $TAG > /dev/null 2>&1; tail -4 gpurun_out/ablation_$TAG.txt | cut -c1-160
for c in C3 C4; do timeout 900 python3 tools/list_stats.py $c 2>&1 | grep -v amdgpu | tail -2; done | tee $O/list_stats.txt
python3 -c "
import json
d=json.load(open('$O/bench_default.json')); print('default:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_whole_step'], d.get('end_to_end_ms'))
d=json.load(open('$O/bench_2ranks_gloo_one_gpu.json')); print('2 ranks:', d['n_gpus'], d['value'], d.get('gather_note'), [ (s['config'], s.get('value'), s.get('skipped')) for s in d['strong_scaling']], d['config']['lineage_sharded_by_genes'])
"
