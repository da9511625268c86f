#!/bin/bash
bash tools/final_round.sh r05m > gpurun_out/final_round_r05m.log 2>&1
mkdir -p gpurun_out/r05m
timeout 2000 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r05m/pytest_gpu_full.txt; cat gpurun_out/r05m/pytest_gpu_full.txt
tail -c 300 gpurun_out/r05m/bench_default.json
