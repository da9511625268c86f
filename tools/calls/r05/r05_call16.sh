#!/bin/bash
mkdir -p gpurun_out/r05g
O=gpurun_out/r05g
rm -f $O/cfg1_group.txt
for g in 3 4 6 8 12 3; do
  NEFII_TRACE_GROUP=$g timeout 600 python3 bench.py --workload cfg1 --steps 480 --warmup 48 --repeats 3 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/c1.json
  python3 -c "import json; d=json.loads(open('$O/c1.json').read()); print('cfg1 trace group $g:', round(d['ms_per_step'],4), [round(x,3) for x in d['ms_per_step_repeats']], d.get('invalid'))" | tee -a $O/cfg1_group.txt
done
for g in 1 2 3 1; do
  NEFII_TRACE_GROUP=$g timeout 600 python3 bench.py --workload cfg2 --steps 240 --warmup 24 --repeats 3 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/c2.json
  python3 -c "import json; d=json.loads(open('$O/c2.json').read()); print('cfg2 trace group $g:', round(d['ms_per_step'],4), [round(x,3) for x in d['ms_per_step_repeats']], d.get('invalid'))" | tee -a $O/cfg1_group.txt
done
