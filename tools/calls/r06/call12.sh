#!/bin/bash
# round 6, last call: the GPU suite, smoke() and the flag-less bench on the final tree (after the forced rebuild)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q --durations=10 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_gpu_full.txt
tail -3 $O/pytest_gpu_full.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
/usr/bin/time -v python3 bench.py --full-out $O/bench_full_default.json > $O/bench_default_stdout.txt 2> $O/bench_default.err
grep "Elapsed (wall" $O/bench_default.err
tail -1 $O/bench_default_stdout.txt > $O/bench_default_compact.json; wc -c $O/bench_default_compact.json; tail -c 700 $O/bench_default_compact.json
