#!/bin/bash
# round 6, call 10: "16f" with its last layer in the fp16 split (lo image aliased over the fp8 images): errors, tile time, budget
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -q -s -k "split_fp8 or fp8corr or cfg3-tier-fp8" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_call10.txt
grep "split_fp8\|fp8corr\|gradients with\|passed\|failed\|^E " $O/pytest_call10.txt | cut -c1-330 | head -40
(for sc in bowl_trained bowl_dense; do MODEL=conf SCENE=$sc timeout 300 python3 tools/eval_microbench.py 12; done; timeout 300 python3 tools/eval_microbench.py 12) 2>&1 | grep -v amdgpu > $O/eval_microbench_fp8.txt
cat $O/eval_microbench_fp8.txt
timeout 1200 python3 tools/error_budget.py $O/error_budget.json 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" > $O/error_budget.txt
cat $O/error_budget.txt
