#!/bin/bash
# the round's closing call: long runs with the staged search, the whole GPU suite, the flag-less bench line
mkdir -p gpurun_out/r05h
O=gpurun_out/r05h
NEFII_MINSDF_STAGED=0 timeout 600 python3 tools/long_train.py cfg3 300 4 $O/long_train_cfg3_staged0.json > $O/long_train_cfg3_staged0.log 2>&1
timeout 600 python3 tools/long_train.py cfg3 300 4 $O/long_train_cfg3_staged1.json > $O/long_train_cfg3_staged1.log 2>&1
timeout 600 python3 tools/long_train.py cfg4 150 4 $O/long_train_cfg4.json > $O/long_train_cfg4.log 2>&1
timeout 600 python3 tools/long_train.py cfg2 1000 4 $O/long_train_cfg2.json > $O/long_train_cfg2.log 2>&1
timeout 600 python3 tools/long_train.py cfg1 2000 4 $O/long_train_cfg1.json > $O/long_train_cfg1.log 2>&1
tail -2 $O/long_train_*.log
timeout 2000 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu_full.txt; cat $O/pytest_gpu_full.txt
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default_time.txt; tail -3 $O/bench_default_time.txt
tail -c 400 $O/bench_default.json
