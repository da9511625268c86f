#!/bin/bash
# PMC passes over tools/eval_microbench.py (one counter group per run) -> per-kernel averages.
# usage: pmc_microbench.sh [tiles_per_cu] [groups...]; NEFII_LIB_PATH selects the build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-8}; shift
S=/tmp/pmcmb; rm -rf $S; mkdir -p $S
if [ $# -eq 0 ]; then set -- "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; fi
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $S/p$i -- python3 tools/eval_microbench.py $T > /dev/null 2> $S/err$i.log
  f=$(find $S/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'sdf_points' in r['Kernel_Name']:
        kind = 'single pass (16s)' if 'kernel16s' in r['Kernel_Name'] else 'split (16q)'
        a = acc[(kind, r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
for (kind, k), (v, n) in sorted(acc.items()):
    print('%-18s %-32s %16.0f  (avg over %d launches)' % (kind, k, v / n, n))
PY
done
