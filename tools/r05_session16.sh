#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s16; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.txt
timeout 900 python3 bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json; python3 -c "
import json; d=json.load(open('$O/bench_default.json'))
print({k: d[k] for k in ('value','ms_per_step','ms_per_step_unchecked','ms_per_step_plan_order','ms_per_step_cold','end_to_end_ms','end_to_end_ms_int32')})
print(d['roofline']['frac'], d['roofline']['frac_whole_step'], d['roofline']['kernel_ms'])
print(d.get('north_star_shape'))"
PROSSTT_BENCH_BACKEND=gloo PROSSTT_BENCH_ONE_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 5 --warmup 2 2> $O/bench2.err | tail -1 > $O/bench_2ranks_gloo_one_gpu.json
python3 -c "
import json; d=json.load(open('$O/bench_2ranks_gloo_one_gpu.json'))
print('2 ranks:', d['n_gpus'], d['value'], d.get('gather_note'), d.get('pipeline_note'), d.get('extras_error'), [(s['config'], s.get('value'), s.get('skipped')) for s in d.get('strong_scaling', [])])"
tail -3 $O/bench2.err
