#!/bin/bash
# Runs on the GPU box: the round's closing measurements.  usage: tools/final_session.sh <tag>
TAG=${1:-r06}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_$TAG; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_bench.sh $TAG > /dev/null 2>&1; grep "under the tracer\|derived" -A2 gpurun_out/prof_$TAG/summary.txt | cut -c1-300
BENCH_EXTRA="--config T32" bash tools/profile_bench.sh ${TAG}_t32 > /dev/null 2>&1; grep "under the tracer\|derived" -A2 gpurun_out/prof_${TAG}_t32/summary.txt | cut -c1-300
python3 bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
PROSSTT_BENCH_BACKEND=gloo PROSSTT_BENCH_ONE_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 5 --warmup 2 2> $O/bench2.err | tail -1 > $O/bench_2ranks_gloo_one_gpu.json
{
echo "# cells presented grouped by mean-tensor row (KBENCH_SORT=1: what simulation.draw_counts does), 20 calls back to back per round:"
for c in C3 T32 C2 C4 C5; do if [ $c = C5 ]; then export KBENCH_CELLS=125000; else unset KBENCH_CELLS; fi; KBENCH_SORT=1 KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py $c 8 shipped build/ab/libprosstt_amd_r5.so 2>&1 | grep -v amdgpu; done
echo "# cells in the order of the plan:"
for c in C3 T32 C2 C4 C5; do if [ $c = C5 ]; then export KBENCH_CELLS=125000; else unset KBENCH_CELLS; fi; KBENCH_BURST=20 timeout 600 python3 tools/kbench_ab.py $c 8 shipped build/ab/libprosstt_amd_r5.so 2>&1 | grep -v amdgpu; done
} | tee $O/other_configs.txt
unset KBENCH_CELLS
KBENCH_SORT=1 bash tools/stage_budget.sh ${TAG}_c3 C3 > /dev/null 2>&1; tail -7 gpurun_out/stage_budget_${TAG}_c3.txt | cut -c1-170
KBENCH_SORT=1 bash tools/stage_budget.sh ${TAG}_t32 T32 > /dev/null 2>&1; tail -7 gpurun_out/stage_budget_${TAG}_t32.txt | cut -c1-170
for c in C3 T32 C4; do timeout 900 python3 tools/list_stats.py $c 2>&1 | grep -v amdgpu | tail -3; done | tee $O/list_stats.txt
KBENCH_SORT=1 KBENCH_BURST=20 rocprofv3 --kernel-trace --output-format csv -d $O/trace_gap -- python3 tools/kbench_ab.py C3 4 shipped > /dev/null 2>&1; python3 tools/gap_trace.py $O/trace_gap | tee $O/gaps_C3.txt; rm -rf $O/trace_gap
timeout 300 python3 tools/cold_probe.py 2>&1 | grep -v amdgpu > $O/cold_probe.txt
# the RCCL side of bench.py on what one GPU allows: a one-rank process group on nccl (init, barrier, the MAX all-reduce of the timing)
PROSSTT_BENCH_FORCE_DIST=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-cells 0 --no-end-to-end --no-target-shape 2> $O/bench_nccl_1rank.err | tail -1 > $O/bench_nccl_1rank.json
python3 -c "
import json; d=json.load(open('$O/bench_nccl_1rank.json')); print('nccl, 1 rank:', d['n_gpus'], d['value'], d['roofline']['frac'])"
for t in c4:"--config C4" c2:"--config C2" c5share:"--config C5 --cells-per-gpu 125000"; do BENCH_EXTRA="${t#*:}" bash tools/profile_bench.sh ${TAG}_${t%%:*} > /dev/null 2>&1; grep "bench_unprofiled.json" gpurun_out/prof_${TAG}_${t%%:*}/summary.txt | cut -c1-200; done
python3 -c "
import json
d=json.load(open('$O/bench_default.json')); print('default:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_whole_step'], d.get('end_to_end_ms'), d['ms_per_step_cold'], d['north_star_shape']['frac'], d['north_star_shape']['frac_whole_step'])
d=json.load(open('$O/bench_2ranks_gloo_one_gpu.json')); print('2 ranks:', d['n_gpus'], d['value'], d.get('gather_note'), d.get('pipeline_note'), [ (s['config'], s.get('value'), s.get('skipped')) for s in d['strong_scaling']], d['config']['lineage_sharded_by_genes'])
"
