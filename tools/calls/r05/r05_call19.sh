#!/bin/bash
mkdir -p gpurun_out/r05i
O=gpurun_out/r05i
rm -f $O/ab_bracket.txt
for w in cfg3 cfg4 cfg2; do
  st=10; [ $w = cfg2 ] && st=200
  for d in 0 1 0 1; do
    NEFII_BRACKET_STAGED=$d timeout 600 python3 bench.py --workload $w --steps $st --warmup 5 --repeats 1 --no-cpu-baseline --no-side-measurement > $O/bench_${w}_b$d.json 2>$O/bench_${w}_b$d.err
    python3 -c "import json,sys; d=json.loads(open('$O/bench_${w}_b$d.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$w bracket staged=$d', round(d['ms_per_step'],4), r.get('frac'), r.get('frac_executed'), r.get('sdf_evals_executed_single_pass'), r.get('sdf_evals_executed_split_precision'), r.get('minsdf_lipschitz_violation'), d.get('invalid'))" | tee -a $O/ab_bracket.txt
  done
done
for d in 0 1 0 1; do
  NEFII_BRACKET_STAGED=$d timeout 600 python3 bench.py --workload cfg5 --frame-rows 32 2>/dev/null | tail -1 > $O/bench_cfg5_b$d.json
  python3 -c "import json; d=json.loads(open('$O/bench_cfg5_b$d.json').read()); print('cfg5 band bracket staged=$d', d.get('ms_per_step'), d.get('value'))" | tee -a $O/ab_bracket.txt
done
