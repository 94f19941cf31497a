#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s13; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -60 > $O/pytest.txt; tail -45 $O/pytest.txt
timeout 900 python3 bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json; python3 -c "
import json; d=json.load(open('$O/bench_default.json'))
print({k: d[k] for k in ('value','ms_per_step','ms_per_step_unchecked','ms_per_step_plan_order','ms_per_step_cold','end_to_end_ms','end_to_end_ms_int32')})
print(d['roofline']['frac'], d['roofline']['frac_whole_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'])
print(d.get('north_star_shape'))"
tail -5 $O/bench.err
