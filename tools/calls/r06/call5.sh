#!/bin/bash
# round 6, call 5: soak of tier + fp8 (long training runs, 4 batches cycled), the fp8 tests with their final bounds, the flag-less bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_longrun.py -m gpu -q -s -k "split_fp8 or cfg3-tier-fp8 or config3_shrunk_trains" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_call5.txt
grep "split_fp8\|longrun\|parity\|worst\|passed\|failed\|^E " $O/pytest_call5.txt | head -40
for w in cfg3 cfg2 cfg1; do
  NEFII_TRACE_TIER=1 NEFII_SPLIT_FP8=1 timeout 900 python3 tools/long_train.py $w $([ $w = cfg3 ] && echo 400 || echo 2000) 4 $O/long_train_${w}_tier_fp8.json 2>/dev/null | tail -1 | cut -c1-600
done
python3 bench.py --full-out $O/bench_full_default.json > $O/bench_default_stdout.txt 2> $O/bench_default.err
tail -1 $O/bench_default_stdout.txt | tee $O/bench_default_compact.json
