#!/bin/bash
# 16d A/B: bit identity, tile microbenchmark, configs 2 / 3
mkdir -p gpurun_out/r05d
O=gpurun_out/r05d
export SCENE=bowl_trained
NEFII_COARSE_D=0 timeout 300 python3 tools/experiments/coarse_d_dump.py $O/v0.npy > $O/dump.txt 2>&1
NEFII_COARSE_D=1 timeout 300 python3 tools/experiments/coarse_d_dump.py $O/v1.npy >> $O/dump.txt 2>&1
python3 - >> $O/dump.txt 2>&1 <<'PY'
import numpy as np
a, b = np.load('gpurun_out/r05d/v0.npy'), np.load('gpurun_out/r05d/v1.npy')
print('bit-identical:', bool((a.view(np.uint32) == b.view(np.uint32)).all()), ' max |diff| %.3e' % np.abs(a - b).max())
PY
rm -f $O/v0.npy $O/v1.npy
cat $O/dump.txt
for d in 0 1 0 1; do
  echo "== NEFII_COARSE_D=$d" >> $O/microbench.txt
  NEFII_COARSE_D=$d timeout 300 python3 tools/eval_microbench.py 1 2 4 12 2>&1 | grep "single pass" >> $O/microbench.txt
done
cat $O/microbench.txt
for d in 0 1 0 1; do
  NEFII_COARSE_D=$d timeout 600 python3 bench.py --workload cfg3 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline > $O/bench_cfg3_d$d.json 2>$O/bench_cfg3_d$d.err
  python3 -c "import json,sys; d=json.loads(open('$O/bench_cfg3_d$d.json').read().strip().splitlines()[-1]); print('cfg3 D=$d', d['ms_per_step'], d['roofline'].get('board_power'))"
  NEFII_COARSE_D=$d timeout 600 python3 bench.py --workload cfg2 --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline > $O/bench_cfg2_d$d.json 2>$O/bench_cfg2_d$d.err
  python3 -c "import json,sys; d=json.loads(open('$O/bench_cfg2_d$d.json').read().strip().splitlines()[-1]); print('cfg2 D=$d', d['ms_per_step'])"
done
