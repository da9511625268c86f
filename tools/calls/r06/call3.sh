#!/bin/bash
# round 6, call 3: the "16f" evaluator (correction products on block-scaled fp8) - correctness, tile time, tracer A/B; the fixed
# adversarial tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -s -x -k "fp8corr" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_call3_fp8.txt
tail -30 $O/pytest_call3_fp8.txt
(for sc in bowl_trained frame_trained bowl_dense; do MODEL=conf SCENE=$sc timeout 300 python3 tools/eval_microbench.py 12; done; timeout 300 python3 tools/eval_microbench.py 1 12) 2>&1 | grep -v amdgpu > $O/eval_microbench_fp8.txt
cat $O/eval_microbench_fp8.txt
timeout 1200 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "split_fp8 or adversarial" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_call3.txt
grep "split_fp8\|adversarial\|passed\|failed\|^E " $O/pytest_call3.txt | head -60
for f in 0 1; do
  NEFII_SPLIT_FP8=$f timeout 600 python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-side-measurement --full-out $O/bench_cfg3_fp8_$f.json 2>/dev/null | tail -1 > $O/bench_cfg3_fp8_${f}_compact.json
done
python3 - <<'PY'
import json
for f in (0,1):
    j=json.load(open('gpurun_out/r06/bench_cfg3_fp8_%d.json'%f)); r=j['roofline']
    print('split_fp8=%d: %.2f ms/step %s kernel ms %.1f frac %.3f issued %.0f'%(f,j['ms_per_step'],j['ms_per_step_repeats'],r['kernel_ms_per_step'],r['frac'],r['issued_tflops']))
PY
