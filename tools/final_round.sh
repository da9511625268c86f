#!/bin/bash
# Everything profiles/rNN/ holds for a round, in one gpurun call:  tools/final_round.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-r05}; O=gpurun_out/$T
mkdir -p $O
for w in cfg2 cfg3 cfg4; do
    bash tools/profile_round.sh $T $w eval_kernel16 $([ $w = cfg2 ] && echo 20 || echo 10) > $O/log_$w.txt 2>&1
done
for w in cfg3 cfg4; do
    rm -rf /tmp/np
    NEFII_BENCH_PREFETCH=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-side-measurement > $O/bench_${w}_noprefetch_under_rocprof.json 2> /dev/null
    cp $(find /tmp/np -name "*kernel_stats.csv" | head -1) $O/bench_${w}_noprefetch_kernel_stats.csv
done
python3 bench.py --workload cfg1 --steps 240 --warmup 48 --no-cpu-baseline > $O/bench_cfg1.json 2> /dev/null
rm -rf /tmp/c1; rocprofv3 --kernel-trace --output-format csv -d /tmp/c1 -- python3 bench.py --workload cfg1 --steps 120 --warmup 48 --no-cpu-baseline --no-side-measurement > /dev/null 2>&1
python3 tools/queue_listing.py $(find /tmp/c1 -name "*kernel_trace.csv" | head -1) 5 > $O/cfg1_tail_listing_after.txt 2>&1
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
(python3 tools/mlp_microbench.py 4096 139264; MODEL=neus python3 tools/mlp_microbench.py 139264; echo "== NEFII_MLP_STREAM=0"
 NEFII_MLP_STREAM=0 python3 tools/mlp_microbench.py 4096 139264; MODEL=neus NEFII_MLP_STREAM=0 python3 tools/mlp_microbench.py 139264
 echo "== NEFII_MLP_H16=0 (fp32 stash and dz)"; NEFII_MLP_H16=0 python3 tools/mlp_microbench.py 4096 139264) 2>&1 | grep -v amdgpu > $O/mlp_microbench.txt
(python3 tools/wgrad_microbench.py; echo "== NEFII_WGRAD_TR=0"; NEFII_WGRAD_TR=0 python3 tools/wgrad_microbench.py) 2>&1 | grep -v amdgpu > $O/wgrad_microbench.txt
(python3 tools/eval_microbench.py 1 3 12; MODEL=neus python3 tools/eval_microbench.py 1 3 12
 for sc in bowl_trained frame_trained bowl_dense bowl; do MODEL=conf SCENE=$sc python3 tools/eval_microbench.py 12; done) 2>&1 | grep -v amdgpu > $O/eval_microbench.txt
python3 tools/render_bench.py 2> /dev/null | tail -1 > $O/render_cfg5_crop.json
python3 tools/trace_rounds.py cfg3 2> /dev/null | grep -v "Warning\|WeightNorm\|amdgpu" > $O/rounds_cfg3.txt
python3 tools/trace_rounds.py cfg2 2> /dev/null | grep -v "Warning\|WeightNorm\|amdgpu" > $O/rounds_cfg2.txt
(bash tools/pmc_microbench.sh 12) > $O/pmc_microbench_eval_tile.txt 2>&1
tools/probes/slot_probe > $O/slot_probe.txt 2>&1
tools/probes/fp8_probe > $O/fp8_probe.txt 2>&1
python3 tools/tier_parity.py 32768 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" > $O/tier_parity_tracer.txt
python3 tools/experiments/grad_probe.py 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" > $O/grad_probe_cfg3.txt
for tier in 0 1; do
    NEFII_TRACE_TIER=$tier python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-side-measurement 2> /dev/null | tail -1 > $O/bench_cfg3_tier$tier.json
    NEFII_TRACE_TIER=$tier python3 bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-side-measurement 2> /dev/null | tail -1 > $O/bench_cfg4_tier$tier.json
done
python3 tools/trace_rounds_secondary.py cfg3 2> /dev/null | grep -v "Warning\|WeightNorm\|amdgpu" > $O/rounds_cfg3_secondary.txt
for st in 0 1; do      # the staged bracket search alone (the staged min-SDF search stays on), same box
    for w in cfg3 cfg4; do
        NEFII_BRACKET_STAGED=$st python3 bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-side-measurement 2> /dev/null | tail -1 > $O/bench_${w}_bracket$st.json
    done
    NEFII_BRACKET_STAGED=$st python3 bench.py --workload cfg5 --frame-rows 32 2> /dev/null | tail -1 > $O/bench_cfg5_bracket$st.json
done
for st in 0 1; do      # both staged searches (NEFII_MINSDF_STAGED switches the slope bound off altogether), same box
    for w in cfg3 cfg4 cfg2; do
        NEFII_MINSDF_STAGED=$st python3 bench.py --workload $w --steps $([ $w = cfg2 ] && echo 200 || echo 10) --warmup 5 --no-cpu-baseline --no-side-measurement 2> /dev/null | tail -1 > $O/bench_${w}_staged$st.json
    done
done
python3 tools/render_full_frame.py $O/render_cfg5 64 > $O/render_cfg5_full_frame.log 2>&1
bash tools/power_probe.sh 2>&1 | grep -v "Warn\|amdgpu.ids" | tr ";" "\n" | grep -v "=====" > $O/power_probe.txt
tail -c 300 $O/bench_default.json
