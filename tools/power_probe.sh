#!/bin/bash
# Board power and clocks (rocm-smi, as far as an ordinary user may read them) while the chip runs (a) a bare fp16 MFMA loop on
# random operands (nefii_mfma_sustained_probe), (b) the split-precision tile evaluator, (c) the single-pass one, (d) nothing.
cd "$GRAFT_REPO_ROOT" || exit 1
poll() { for i in 1 2 3 4 5 6; do sleep 0.5; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|junction" | tr '\n' ';'; echo; done; }
mkdir -p gpurun_out/r04; echo "== power cap"; rocm-smi --showmaxpower 2>/dev/null | grep -i "power"; echo "== idle"; poll | tail -2
python3 - <<'PY' &
import ctypes, sys
sys.path.insert(0, '.')
import torch
from nefii_amd import _lib
lib = _lib.lib(); ms, fl = ctypes.c_float(), ctypes.c_double()
for _ in range(14): lib.nefii_mfma_sustained_probe(2500000, ctypes.byref(ms), ctypes.byref(fl), None)
print('probe: %.0f TFLOP/s over %.0f ms launches' % (fl.value / ms.value / 1e9, ms.value))
PY
sleep 3; echo "== bare MFMA loop, random operands"; poll; wait
python3 - <<'PY' &
import sys
sys.path.insert(0, '.')
import torch
from nefii_amd import ops, synthetic as syn
from oracle import nets
mc = syn.model_conf('physg'); sd = syn.make_state_dict(mc, seed=0, bumpy=0.004)
specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
x = (torch.randn(48 * 256 * 64, 3) * 0.4).cuda()
import time
for coarse in (False, True):
    t0 = time.time(); n = 0
    while time.time() - t0 < 5.0:
        for _ in range(20): ops.sdf_eval(pm, x, coarse=coarse)
        torch.cuda.synchronize(); n += 20
    print('%s: %.3f ms per launch of %d points' % ('single pass' if coarse else 'split', (time.time() - t0) / n * 1e3, x.shape[0]), flush=True)
PY
sleep 4; echo "== split evaluator (first 5 s), then single pass"; poll; poll; wait
