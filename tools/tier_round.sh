#!/bin/bash
# The tiered sphere tracing's parity protocol and A/B in one gpurun call:  tools/tier_round.sh <tag>
#   A  tools/tier_parity.py: the tracer alone, tier off / on / oracle
#   B  the GPU suite's oracle and golden comparisons with NEFII_TRACE_TIER=1 forced for every batch size
#   C  bench.py --workload cfg3 / cfg4 with the tier off and on, alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-r05}; O=gpurun_out/$T
mkdir -p $O
python3 tools/tier_parity.py 32768 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" > $O/tier_parity_tracer.txt
# (--parity-soft: tests/parity.py prints a bound that does not hold instead of raising - the run lists every figure)
for tier in 0 1; do
    NEFII_TRACE_TIER=$tier timeout 1500 python3 -m pytest --parity-soft tests/test_gpu_configs.py tests/test_gpu_longrun.py \
        tests/test_gpu_renderer.py -m gpu -q -s -k "config or longrun or long or golden or full_size or indirect" \
        2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/tier_parity_suite_tier$tier.txt
done
for rep in 1 2; do
    for tier in 0 1; do
        NEFII_TRACE_TIER=$tier python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-side-measurement \
            2> /dev/null | tail -1 > $O/bench_cfg3_tier${tier}_rep$rep.json
    done
done
for tier in 0 1; do
    NEFII_TRACE_TIER=$tier python3 bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-side-measurement \
        2> /dev/null | tail -1 > $O/bench_cfg4_tier${tier}.json
    NEFII_TRACE_TIER=$tier python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-side-measurement \
        2> /dev/null | tail -1 > $O/bench_cfg2_tier${tier}.json
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/r05/bench_cfg*_tier*.json')):
    try:
        j = json.load(open(f))
        r = j['roofline']
        print('%-50s %8.2f ms/step  frac %.3f executed %.3f  split %d single-pass %d  tier queries %d repeated %d  kernel ms %.1f' % (
            f.split('/')[-1], j['ms_per_step'], r['frac'], r['frac_executed'], r['sdf_evals_executed_split_precision'],
            r['sdf_evals_executed_single_pass'], r['tier_queries_single_pass'], r['tier_queries_repeated'], r['kernel_ms_per_step']))
    except Exception as e:
        print(f, 'unreadable', e)
PY
