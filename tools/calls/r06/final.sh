#!/bin/bash
# round 6, closing call: everything profiles/r06/ holds that is not there yet
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q --durations=20 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/pytest_gpu_full.txt
tail -4 $O/pytest_gpu_full.txt
python3 bench.py --full-out $O/bench_full_default.json > $O/bench_default_stdout.txt 2> $O/bench_default.err
tail -1 $O/bench_default_stdout.txt > $O/bench_default_compact.json; wc -c $O/bench_default_compact.json
bash tools/profile_round.sh r06 cfg3 eval_kernel16 10 > $O/log_cfg3.txt 2>&1
rm -rf /tmp/np
NEFII_BENCH_PREFETCH=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np -- python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-side-measurement --full-out $O/bench_cfg3_noprefetch_under_rocprof.json > /dev/null 2> /dev/null
cp $(find /tmp/np -name "*kernel_stats.csv" | head -1) $O/bench_cfg3_noprefetch_kernel_stats.csv
python3 tools/kernel_stats_per_step.py $O/bench_cfg3_noprefetch_kernel_stats.csv $O/bench_cfg3_noprefetch_per_step.txt | head -16
for w in cfg4 cfg2 cfg1; do
  python3 bench.py --workload $w --steps $([ $w = cfg4 ] && echo 10 || echo 200) --warmup $([ $w = cfg4 ] && echo 3 || echo 48) --no-cpu-baseline --full-out $O/bench_$w.json 2>/dev/null | tail -1 > $O/bench_${w}_compact.json
done
python3 tools/render_full_frame.py $O/render_cfg5 64 > $O/render_cfg5_full_frame.log 2>&1
cp $O/render_cfg5/render_cfg5_full_frame.json $O/ 2>/dev/null; tail -3 $O/render_cfg5_full_frame.log
rm -rf $O/render_cfg5
tail -c 600 $O/bench_default_compact.json
