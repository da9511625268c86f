#!/bin/bash
mkdir -p gpurun_out/r05d
O=gpurun_out/r05d
export SCENE=bowl_trained
rm -f $O/dump4.txt $O/microbench4.txt $O/ab4.txt
for d in 0 1; do
  NEFII_COARSE_D=$d timeout 300 python3 tools/experiments/coarse_d_dump.py $O/v$d.npy >> $O/dump4.txt 2>&1
done
python3 - >> $O/dump4.txt 2>&1 <<'PY'
import numpy as np
a = np.load('gpurun_out/r05d/v0.npy')
b = np.load('gpurun_out/r05d/v1.npy')
print('D=1 bit-identical to 16s:', bool((a.view(np.uint32) == b.view(np.uint32)).all()), ' max |diff| %.3e' % np.abs(a - b).max())
PY
rm -f $O/v?.npy
grep -v amdgpu.ids $O/dump4.txt
for d in 0 1; do
  echo "== NEFII_COARSE_D=$d" >> $O/microbench4.txt
  NEFII_COARSE_D=$d timeout 300 python3 tools/eval_microbench.py 1 2 4 12 24 2>&1 | grep "single pass" >> $O/microbench4.txt
done
cat $O/microbench4.txt
for w in cfg3 cfg2 cfg1; do
  st=10; [ $w = cfg2 ] && st=200; [ $w = cfg1 ] && st=400
  for d in "0 256" "1 256" "1 512" "0 256" "1 256" "1 512"; do
    set -- $d
    NEFII_COARSE_D=$1 NEFII_COARSE_D_GRID=$2 timeout 600 python3 bench.py --workload $w --steps $st --warmup 5 --repeats 1 --no-cpu-baseline > $O/bench_${w}_d$1.json 2>$O/bench_${w}_d$1.err
    python3 -c "import json,sys; d=json.loads(open('$O/bench_${w}_d$1.json').read().strip().splitlines()[-1]); print('$w D=$1 grid $2', round(d['ms_per_step'],4), d['roofline'].get('board_power', {}).get('avg_w'))" | tee -a $O/ab4.txt
  done
done
NEFII_COARSE_D=1 timeout 1200 python3 -m pytest tests/test_gpu_concurrency.py -q -x 2>&1 | tail -5 | tee $O/pytest_concurrency.txt
