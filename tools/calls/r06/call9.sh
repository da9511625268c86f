#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q --durations=15 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_gpu_full.txt
tail -4 $O/pytest_gpu_full.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
