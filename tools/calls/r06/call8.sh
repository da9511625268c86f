#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_renderer.py -m gpu -q -s -k "adversarial or (recovers_a_rendered_target and physg)" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_call8.txt
grep "adversarial\|passed\|failed\|^E " $O/pytest_call8.txt | cut -c1-300 | head -40
