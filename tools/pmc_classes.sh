#!/bin/bash
# Per-class SQ instruction counters of the tile evaluators (VERDICT r3 next #1a): which resource fills the layer period?
# One rocprofv3 --pmc pass per counter group over tools/eval_microbench.py; names absent from `rocprofv3 -L` are dropped.
# usage: pmc_classes.sh [tiles_per_cu] ; output: per-kernel averages on stdout (commit under profiles/rNN/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-12}
S=/tmp/pmccls; rm -rf $S; mkdir -p $S
rocprofv3 -L > $S/avail.txt 2>&1
GROUPS_=(
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_SALU"
 "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT"
 "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES GRBM_GUI_ACTIVE"
)
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1)); keep=""
  for c in $grp; do if grep -qw "$c" $S/avail.txt; then keep="$keep $c"; else echo "# not available on this device: $c"; fi; done
  [ -z "$keep" ] && continue
  rocprofv3 --pmc $keep --output-format csv -d $S/p$i -- python3 tools/eval_microbench.py $T > $S/out$i.log 2> $S/err$i.log
  f=$(find $S/p$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "# group $i produced no counter file:"; tail -5 $S/err$i.log; continue; fi
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r['Kernel_Name']
    if 'sdf_points' in kn:
        kind = 'single pass (16s)' if 'kernel16s' in kn else ('coarse-x (16c)' if 'kernel16c' in kn else 'split (16q)')
        a = acc[(kind, r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
for (kind, k), (v, n) in sorted(acc.items()):
    print('%-18s %-32s %18.0f  (avg over %d launches)' % (kind, k, v / n, n))
PY
done
