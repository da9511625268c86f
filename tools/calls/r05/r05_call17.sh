#!/bin/bash
mkdir -p gpurun_out/r05g
O=gpurun_out/r05g
timeout 1500 python3 -m pytest tests/test_gpu_renderer.py tests/test_gpu_longrun.py tests/test_gpu_concurrency.py -q -x 2>&1 | tail -6 | tee $O/pytest_group8.txt
python3 bench.py --workload cfg1 --steps 240 --warmup 48 --no-cpu-baseline > $O/bench_cfg1.json 2>/dev/null
python3 -c "import json; d=json.loads(open('$O/bench_cfg1.json').read().strip().splitlines()[-1]); print('cfg1', d['ms_per_step'], d['ms_per_step_repeats'], d['value'], d.get('invalid'), d['roofline'].get('frac'))"
