#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "tier" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_call7.txt
grep "gradients with\|worst\|passed\|failed\|^E " $O/pytest_call7.txt | head -40
