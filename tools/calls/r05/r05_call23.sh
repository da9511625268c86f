#!/bin/bash
mkdir -p gpurun_out/r05o
O=gpurun_out/r05o
rm -f $O/cfg2_group.txt
for g in 1 3 4 6 1 3; do
  NEFII_TRACE_GROUP=$g timeout 600 python3 bench.py --workload cfg2 --steps 240 --warmup 36 --repeats 3 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/c2.json
  python3 -c "import json; d=json.loads(open('$O/c2.json').read()); print('cfg2 trace group $g:', round(d['ms_per_step'],4), [round(x,3) for x in d['ms_per_step_repeats']], d.get('invalid'))" | tee -a $O/cfg2_group.txt
done
