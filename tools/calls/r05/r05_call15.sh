#!/bin/bash
mkdir -p gpurun_out/r05g
O=gpurun_out/r05g
for mr in 1024 256 1024 256; do
  NEFII_COARSE_MIN_RAYS=$mr timeout 600 python3 bench.py --workload cfg1 --steps 400 --warmup 20 --repeats 3 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/c1.json
  python3 -c "import json; d=json.loads(open('$O/c1.json').read()); print('cfg1 coarse_min_rays $mr:', round(d['ms_per_step'],4), [round(x,3) for x in d['ms_per_step_repeats']], d.get('invalid'))" | tee -a $O/cfg1_coarse.txt
done
