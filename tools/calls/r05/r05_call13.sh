#!/bin/bash
mkdir -p gpurun_out/r05e
O=gpurun_out/r05e
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py -q -x -s -k "staged_min_sdf or coarse_pass_changes_no_decision or tracer_golden or tracer_audits" 2>&1 | grep -v "^$" | tail -40 | tee $O/pytest_staged.txt
for w in cfg3 cfg2 cfg4; do
  st=10; [ $w = cfg2 ] && st=200
  for d in 0 1 0 1; do
    NEFII_MINSDF_STAGED=$d timeout 600 python3 bench.py --workload $w --steps $st --warmup 5 --repeats 1 --no-cpu-baseline > $O/bench_${w}_s$d.json 2>$O/bench_${w}_s$d.err
    python3 -c "import json,sys; d=json.loads(open('$O/bench_${w}_s$d.json').read().strip().splitlines()[-1]); print('$w staged=$d', round(d['ms_per_step'],4), d['roofline'].get('frac'), d['roofline'].get('frac_executed'), d.get('invalid'))" | tee -a $O/ab_staged.txt
  done
done
