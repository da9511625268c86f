#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
NEFII_TRACE_TIER=1 NEFII_SPLIT_FP8=1 timeout 900 python3 tools/long_train.py cfg3 300 4 $O/long_train_cfg3_tier_fp8.json 2>/dev/null | tail -1 | cut -c1-400
NEFII_TRACE_TIER=1 NEFII_SPLIT_FP8=1 timeout 900 python3 tools/long_train.py cfg2 2000 4 $O/long_train_cfg2_tier_fp8.json 2>/dev/null | tail -1 | cut -c1-300
timeout 900 python3 -m pytest tests/test_gpu_longrun.py tests/test_lib_abi.py -m gpu -q -s -k "config3_shrunk or abi" 2>&1 | grep "longrun\|passed\|failed"
