#!/bin/bash
mkdir -p gpurun_out/r05r
O=gpurun_out/r05r
timeout 600 python3 tools/long_train.py cfg3 600 4 $O/long_train_cfg3_final.json > $O/lt3.log 2>&1
timeout 600 python3 tools/long_train.py cfg4 300 4 $O/long_train_cfg4_final.json > $O/lt4.log 2>&1
timeout 600 python3 tools/long_train.py cfg2 3000 4 $O/long_train_cfg2_final.json > $O/lt2.log 2>&1
timeout 600 python3 tools/long_train.py cfg1 5000 4 $O/long_train_cfg1_final.json > $O/lt1.log 2>&1
python3 - <<'PY'
import json
for w in ('cfg3','cfg4','cfg2','cfg1'):
    d=json.load(open('gpurun_out/r05r/long_train_%s_final.json'%w))
    l=d['loss_mean_per_25_steps']
    print(w, d['steps'], 'cancelled', d['steps_cancelled_by_the_nan_guard'], 'finite', d['all_losses_finite'], 'loss', round(l[0],3), '->', round(l[-1],3), 'ms/step', round(d['ms_per_step'],3), 'lookahead', d.get('trace_lookahead'))
PY
