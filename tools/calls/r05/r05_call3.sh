cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r05; mkdir -p $O
F='Warning\|WeightNorm\|amdgpu\|warnings.warn'
timeout 1500 python3 -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "shrunk_in_pixels" 2>&1 | grep -v "$F" > $O/pytest_configs_shrunk.txt
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "half_state_bias or coarse_bound_holds or tiered" 2>&1 | grep -v "$F" > $O/pytest_kernels_new.txt
timeout 1500 python3 -m pytest tests/test_gpu_renderer.py -m gpu -q -s -k "eight_processes" 2>&1 | grep -v "$F" > $O/pytest_eight.txt
python3 tools/experiments/grad_probe.py 2>&1 | grep -v "$F" > $O/grad_probe_cfg3.txt
rm -rf /tmp/c1; rocprofv3 --kernel-trace --output-format csv -d /tmp/c1 -- python3 bench.py --workload cfg1 --steps 60 --warmup 15 --no-cpu-baseline --no-side-measurement > $O/bench_cfg1_under_rocprof.json 2>/dev/null
python3 tools/queue_listing.py $(find /tmp/c1 -name "*kernel_trace.csv" | head -1) 5 > $O/cfg1_tail_listing.txt 2>&1
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/pytest_configs_shrunk.txt $O/pytest_kernels_new.txt $O/pytest_eight.txt; tail -c 600 $O/bench_default.json
