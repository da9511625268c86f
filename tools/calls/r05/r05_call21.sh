#!/bin/bash
mkdir -p gpurun_out/r05k
O=gpurun_out/r05k
rm -f $O/ab_missargmin.txt
for w in cfg3 cfg4; do
  for d in 0 1 0 1; do
    NEFII_X_SKIP_MISS_ARGMIN=$d timeout 600 python3 bench.py --workload $w --steps 10 --warmup 5 --repeats 1 --no-cpu-baseline --no-side-measurement > $O/b.json 2>/dev/null
    python3 -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$w skip-miss-argmin=$d', round(d['ms_per_step'],4), r.get('sdf_evals_executed_single_pass'), r.get('sdf_evals_executed_split_precision'), d.get('invalid'))" | tee -a $O/ab_missargmin.txt
  done
done
for d in 0 1; do
  NEFII_X_SKIP_MISS_ARGMIN=$d timeout 600 python3 bench.py --workload cfg5 --frame-rows 32 2>/dev/null | tail -1 > $O/b5.json
  python3 -c "import json; d=json.loads(open('$O/b5.json').read()); print('cfg5 band skip-miss-argmin=$d', d.get('ms_per_step'))" | tee -a $O/ab_missargmin.txt
done
