#!/bin/bash
# Round-end measurement: bench line, rocprofv3 kernel stats of the same command, and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; separate runs) -> gpurun_out/<tag>/ ; copy what should be judged into profiles/.
#   tools/profile_round.sh <tag> <workload> [kernel substring] [steps]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r02}; WL=${2:-cfg2}; KERN=${3:-eval_kernel16}; STEPS=${4:-20}
O=gpurun_out/$TAG; S=/tmp/prof_${TAG}_$WL; rm -rf $S
mkdir -p $O $S
python3 bench.py --workload $WL --steps $STEPS --warmup 3 > $O/bench_$WL.json 2> $O/bench_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $S/ks -- python3 bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline --no-side-measurement > $O/bench_${WL}_under_rocprof.json 2> $O/ks_err.log
cp $(find $S/ks -name "*kernel_stats.csv" | head -1) $O/bench_${WL}_kernel_stats.csv
# per-step figures with the divisor COUNTED from the trace (calls of idr_loss_kernel), never assumed
python3 tools/kernel_stats_per_step.py $O/bench_${WL}_kernel_stats.csv $O/bench_${WL}_kernel_stats_per_step.txt > /dev/null 2>&1
PS=$(( STEPS < 5 ? STEPS : 5 ))
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $S/pf -- python3 bench.py --workload $WL --steps $PS --warmup 2 --no-cpu-baseline --no-side-measurement > /dev/null 2> $O/pf_err.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $S/pw -- python3 bench.py --workload $WL --steps $PS --warmup 2 --no-cpu-baseline --no-side-measurement > /dev/null 2> $O/pw_err.log
FC=$(find $S/pf -name "*counter_collection.csv" | head -1); WC=$(find $S/pw -name "*counter_collection.csv" | head -1)
echo "counter files: $FC $WC"
python3 tools/pmc_traffic.py "$FC" "$WC" $KERN $O/pmc_traffic_$WL.json
for f in $O/*_err.log; do tail -n 2 $f; done
head -14 $O/bench_${WL}_kernel_stats.csv
