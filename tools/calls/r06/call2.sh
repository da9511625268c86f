#!/bin/bash
# round 6, call 2: adversarial staged-search tests (ABI 15: margin probes, counter 13), the foreign-kernel-beside-traces test, the
# split-vs-mix fp8 probe, the error budget table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_concurrency.py tests/test_gpu_renderer.py tests/test_lib_abi.py -m gpu -q -s \
  -k "adversarial or foreign_reduction or checkpoint_layout or staged_min_sdf or abi" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_call2.txt
tail -25 $O/pytest_call2.txt
tools/probes/fp8_probe > $O/fp8_probe.txt 2>&1
tail -6 $O/fp8_probe.txt
timeout 900 python3 tools/error_budget.py $O/error_budget.json 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" > $O/error_budget.txt
cat $O/error_budget.txt
