#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
T0=$(date +%s)
python3 bench.py --full-out $O/bench_full_default.json > $O/bench_default_stdout.txt 2> $O/bench_default.err
echo "bench.py wall seconds: $(( $(date +%s) - T0 )), rc $?"
tail -1 $O/bench_default_stdout.txt > $O/bench_default_compact.json; wc -c $O/bench_default_compact.json; cat $O/bench_default_compact.json
