#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python3 tools/error_budget.py $O/error_budget.json 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" > $O/error_budget.txt
cat $O/error_budget.txt
