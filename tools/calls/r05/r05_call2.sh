cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r05; mkdir -p $O
tools/probes/fp8_probe > $O/fp8_probe.txt 2>&1
for s in "bowl conf" "frame conf" "bowl neus"; do
  timeout 1200 python3 tools/train_scene_sdf.py $s 20000 $O 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" > $O/train_scene_$(echo $s | tr ' ' '_').txt
done
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -q -s -k "half_state_bias or cfg3-bowl" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" | tail -30 > $O/pytest_new_tests.txt
timeout 1500 python3 -m pytest tests/test_gpu_renderer.py -m gpu -q -s -k "eight_processes or min_sdf_on_reporting" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" | tail -30 > $O/pytest_eight.txt
python3 tools/experiments/grad_probe.py 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warn" > $O/grad_probe_cfg3.txt
NEFII_PARITY_SOFT=1 NEFII_TRACE_TIER=1 timeout 1500 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_longrun.py tests/test_gpu_renderer.py -m gpu -q -s -k "config or longrun or long or golden or full_size or indirect" 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu\|warnings.warn" > $O/tier_parity_suite_tier1.txt
tail -3 $O/train_scene_*.txt $O/pytest_new_tests.txt $O/pytest_eight.txt $O/tier_parity_suite_tier1.txt
