#!/bin/bash
mkdir -p gpurun_out/r05q
O=gpurun_out/r05q
rm -f $O/tier_small.txt
for t in 0 1 0 1; do
  NEFII_TRACE_TIER=$t timeout 600 python3 bench.py --workload cfg2 --steps 240 --warmup 36 --repeats 3 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/c.json
  python3 -c "import json; d=json.loads(open('$O/c.json').read()); print('cfg2 tier forced=$t:', round(d['ms_per_step'],4), [round(x,3) for x in d['ms_per_step_repeats']])" | tee -a $O/tier_small.txt
done
for t in 0 1 0 1; do
  NEFII_TRACE_TIER=$t timeout 600 python3 bench.py --workload cfg1 --steps 480 --warmup 64 --repeats 3 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/c.json
  python3 -c "import json; d=json.loads(open('$O/c.json').read()); print('cfg1 tier forced=$t:', round(d['ms_per_step'],4), [round(x,3) for x in d['ms_per_step_repeats']])" | tee -a $O/tier_small.txt
done
