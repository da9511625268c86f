cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05; mkdir -p $O
for tier in 1 0; do
  NEFII_TRACE_TIER=$tier python3 bench.py --workload cfg3 --steps 3 --warmup 2 --repeats 1 --no-side-measurement 2>/dev/null | tail -1 > $O/parity_cfg3_tier$tier.json
done
python3 - <<'PY'
import json
for t in (1,0):
    j=json.load(open('gpurun_out/r05/parity_cfg3_tier%d.json'%t)); print('tier',t, j['parity_vs_cpu_oracle'])
PY
