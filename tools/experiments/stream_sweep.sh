#!/bin/bash
# ms per step of config 1 / 2 against the number of trace streams and HIP's hardware-queue limit (GPU_MAX_HW_QUEUES, default 4)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { # hwq streams workload
  if [ "$1" = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$1; fi
  line=$(NEFII_TRACE_STREAMS=$2 python bench.py --workload $3 --steps 30 --warmup 12 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1)
  echo "hwq=$1 streams=$2 $3 $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(round(d['ms_per_step'],3), [round(x,3) for x in d['ms_per_step_repeats']])" "$line")"
}
for q in ${HWQS:-unset 4 6}; do for s in ${STREAMS:-3 4 5 6}; do for w in cfg1 cfg2; do run $q $s $w; done; done; done
