#!/bin/bash
# round 6, call 4: the 16f evaluator at tracer / config level, the error budget table with its row
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -q -s -k "split_fp8 or fp8corr or cfg3-tier" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_call4.txt
grep "split_fp8\|fp8corr\|parity\|worst\|passed\|failed\|^E " $O/pytest_call4.txt | head -60
timeout 1200 python3 tools/error_budget.py $O/error_budget.json 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" > $O/error_budget.txt
cat $O/error_budget.txt
