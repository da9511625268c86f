#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_concurrency.py tests/test_gpu_renderer.py tests/test_lib_abi.py -m gpu -q -s \
  -k "adversarial or foreign_reduction or checkpoint_layout or staged_min_sdf or abi" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_call2.txt
tail -40 $O/pytest_call2.txt
