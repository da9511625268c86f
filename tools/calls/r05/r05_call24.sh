#!/bin/bash
mkdir -p gpurun_out/r05o
O=gpurun_out/r05o
timeout 1200 python3 -m pytest tests/test_gpu_renderer.py tests/test_gpu_concurrency.py -q -x 2>&1 | tail -4 | tee $O/pytest_group4.txt
python3 bench.py --workload cfg2 --steps 240 --warmup 36 --no-cpu-baseline > $O/bench_cfg2.json 2>/dev/null
python3 -c "import json; d=json.loads(open('$O/bench_cfg2.json').read().strip().splitlines()[-1]); print('cfg2', d['ms_per_step'], d['ms_per_step_repeats'], d['value'], d.get('invalid'), d['roofline'].get('frac'), d['roofline'].get('frac_executed'))"
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/time.txt; tail -3 $O/time.txt
python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('cfg3', d['ms_per_step'], d['ms_per_step_repeats'])
for k in ('cfg1','cfg2','cfg2_near','cfg4','cfg5'):
    v=d.get(k); print(k, v.get('ms_per_step'), v.get('ms_per_step_repeats'))
"
