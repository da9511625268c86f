cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r05; mkdir -p $O
F='Warning\|WeightNorm\|amdgpu\|warnings.warn'
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -k "material_head or prepare_hits or batched_weight or half_state" 2>&1 | grep -v "$F" | tail -5
run() { # tag env...
  tag=$1; shift
  for w in cfg1 cfg2; do
    env "$@" python3 bench.py --workload $w --steps 40 --warmup 15 --no-cpu-baseline --no-side-measurement 2>/dev/null | tail -1 > $O/ab_${w}_$tag.json
  done
}
run all_on NEFII_X=1
run side_off NEFII_RADIANCE_SIDE=0
run all_off NEFII_RADIANCE_SIDE=0 NEFII_WGRAD_BATCH=0 NEFII_PREPARE_HITS=0 NEFII_MATERIAL_HEAD=0
run only_wgrad NEFII_RADIANCE_SIDE=0 NEFII_PREPARE_HITS=0 NEFII_MATERIAL_HEAD=0
run only_prepare NEFII_RADIANCE_SIDE=0 NEFII_WGRAD_BATCH=0 NEFII_MATERIAL_HEAD=0
run only_head NEFII_RADIANCE_SIDE=0 NEFII_WGRAD_BATCH=0 NEFII_PREPARE_HITS=0
run side_off_again NEFII_RADIANCE_SIDE=0
run all_off_again NEFII_RADIANCE_SIDE=0 NEFII_WGRAD_BATCH=0 NEFII_PREPARE_HITS=0 NEFII_MATERIAL_HEAD=0
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05/ab_cfg*.json')):
    try:
        j=json.load(open(f)); print('%-40s %.3f ms %s' % (f.split('/')[-1], j['ms_per_step'], ['%.3f'%x for x in j['ms_per_step_repeats']]))
    except Exception as e: print(f, e)
PY
