#!/bin/bash
# 16d / 16e A/B: bit identity, tile microbenchmark, config 3
mkdir -p gpurun_out/r05d
O=gpurun_out/r05d
export SCENE=bowl_trained
rm -f $O/dump2.txt $O/microbench2.txt
for d in 0 1 2; do
  NEFII_COARSE_D=$d timeout 300 python3 tools/experiments/coarse_d_dump.py $O/v$d.npy >> $O/dump2.txt 2>&1
done
python3 - >> $O/dump2.txt 2>&1 <<'PY'
import numpy as np
a = np.load('gpurun_out/r05d/v0.npy')
for d in (1, 2):
    b = np.load('gpurun_out/r05d/v%d.npy' % d)
    print('D=%d bit-identical to 16s:' % d, bool((a.view(np.uint32) == b.view(np.uint32)).all()), ' max |diff| %.3e' % np.abs(a - b).max())
PY
rm -f $O/v?.npy
grep -v amdgpu.ids $O/dump2.txt
for d in 0 1 2 0 1 2; do
  echo "== NEFII_COARSE_D=$d" >> $O/microbench2.txt
  NEFII_COARSE_D=$d timeout 300 python3 tools/eval_microbench.py 1 2 4 12 24 2>&1 | grep "single pass" >> $O/microbench2.txt
done
cat $O/microbench2.txt
for d in 0 2 1 0 2 1; do
  NEFII_COARSE_D=$d timeout 600 python3 bench.py --workload cfg3 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline > $O/bench_cfg3_d$d.json 2>$O/bench_cfg3_d$d.err
  python3 -c "import json,sys; d=json.loads(open('$O/bench_cfg3_d$d.json').read().strip().splitlines()[-1]); print('cfg3 D=$d', d['ms_per_step'], d['roofline'].get('board_power', {}).get('avg_w'))"
done
