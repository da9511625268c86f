#!/bin/bash
# round 6, call 1: the GPU suite on the new tier rule / compact bench line, then the flag-less bench (compact line + full record)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -q --durations=25 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_gpu_call1.txt
tail -5 $O/pytest_gpu_call1.txt
python3 bench.py --full-out $O/bench_full_default.json > $O/bench_default_stdout.txt 2> $O/bench_default.err
tail -1 $O/bench_default_stdout.txt | tee $O/bench_default_compact.json | wc -c
tail -1 $O/bench_default_stdout.txt
