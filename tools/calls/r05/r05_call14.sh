#!/bin/bash
mkdir -p gpurun_out/r05g
O=gpurun_out/r05g
for la in "3 3" "4 3" "4 4" "6 4" "2 3" "3 3"; do
  set -- $la
  NEFII_BENCH_LOOKAHEAD=$1 NEFII_TRACE_STREAMS=$2 timeout 600 python3 bench.py --workload cfg3 --steps 10 --warmup 5 --repeats 1 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/la.json
  python3 -c "import json; d=json.loads(open('$O/la.json').read()); print('cfg3 lookahead $1 streams $2:', round(d['ms_per_step'],2))" | tee -a $O/lookahead.txt
done
for la in "3 3" "4 4" "6 4" "3 3"; do
  set -- $la
  NEFII_BENCH_LOOKAHEAD=$1 NEFII_TRACE_STREAMS=$2 timeout 600 python3 bench.py --workload cfg4 --steps 10 --warmup 5 --repeats 1 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/la.json
  python3 -c "import json; d=json.loads(open('$O/la.json').read()); print('cfg4 lookahead $1 streams $2:', round(d['ms_per_step'],2))" | tee -a $O/lookahead.txt
done
timeout 900 python3 tools/render_full_frame.py $O/render_cfg5 64 > $O/render_cfg5_full_frame.log 2>&1; tail -3 $O/render_cfg5_full_frame.log
rm -f $O/render_cfg5/*.exr $O/render_cfg5/*.png
