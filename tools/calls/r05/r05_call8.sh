cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r05; mkdir -p $O
F='Warning\|WeightNorm\|amdgpu\|warnings.warn'
(python3 tools/experiments/trace_only.py cfg1 900; python3 tools/experiments/trace_only.py cfg2 300) 2>&1 | grep -v "$F" > $O/trace_only_cfg1_cfg2.txt
for tier in 1 0; do NEFII_TRACE_TIER=$tier python3 tools/long_train.py cfg3 300 4 $O/long_train_cfg3_tier$tier.json 2>&1 | grep -v "$F" | tail -3; done
python3 tools/long_train.py cfg4 150 4 $O/long_train_cfg4.json 2>&1 | grep -v "$F" | tail -2
python3 tools/long_train.py cfg2 1000 4 $O/long_train_cfg2.json 2>&1 | grep -v "$F" | tail -2
python3 tools/render_full_frame.py $O/render_cfg5 64 2>&1 | grep -v "$F" | tail -12 > $O/render_cfg5_full_frame.txt
python3 tools/trace_rounds.py cfg3 2>/dev/null | grep -v "$F" > $O/rounds_cfg3.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/trace_only_cfg1_cfg2.txt; tail -4 $O/render_cfg5_full_frame.txt; tail -c 900 $O/bench_default.json
