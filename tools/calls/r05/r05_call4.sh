cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r05; mkdir -p $O
F='Warning\|WeightNorm\|amdgpu\|warnings.warn'
timeout 2400 python3 -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -40 > $O/pytest_gpu_full.txt
rm -rf /tmp/c1; rocprofv3 --kernel-trace --output-format csv -d /tmp/c1 -- python3 bench.py --workload cfg1 --steps 60 --warmup 15 --no-cpu-baseline --no-side-measurement > $O/bench_cfg1_under_rocprof.json 2>/dev/null
python3 tools/queue_listing.py $(find /tmp/c1 -name "*kernel_trace.csv" | head -1) 5 > $O/cfg1_tail_listing_after.txt 2>&1
for w in cfg1 cfg2; do
  python3 bench.py --workload $w --steps 40 --warmup 15 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/bench_${w}_fused_tail.json
  NEFII_WGRAD_BATCH=0 NEFII_PREPARE_HITS=0 NEFII_RADIANCE_SIDE=0 python3 bench.py --workload $w --steps 40 --warmup 15 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/bench_${w}_unfused_tail.json
done
python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/bench_cfg3_now.json
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05/bench_cfg[123]_*tail.json')+['gpurun_out/r05/bench_cfg3_now.json']):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], '%.3f ms' % j['ms_per_step'], ['%.3f'%x for x in j['ms_per_step_repeats']])
    except Exception as e: print(f, e)
PY
tail -5 $O/pytest_gpu_full.txt; tail -2 $O/cfg1_tail_listing_after.txt
