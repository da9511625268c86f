"""s_memtime stamps at the layer boundaries of the register-resident coarse evaluator (stamp build:
tools/ab_build.sh stamps -DNEFII_C_STAMPS; NEFII_LIB_PATH=build_ab/libnefii_stamps.so python tools/coarse_x_stamps.py)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import _lib, ops, synthetic as syn
from oracle import nets

mc = syn.model_conf('physg')
sd = syn.make_state_dict(mc, seed=0, bumpy=0.004)
specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
n = 12 * 256 * 64
x = (torch.randn(n, 3, generator=torch.Generator().manual_seed(1)) * 0.4).cuda()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 64)()
for _ in range(3):
    ops.sdf_eval(pm, x, coarse=True)
torch.cuda.synchronize()
lib.nefii_debug_c_stamps(buf)
for rep in range(3):
    ops.sdf_eval(pm, x, coarse=True)
    torch.cuda.synchronize()
    lib.nefii_debug_c_stamps(buf)
    for w in range(4):
        t = [buf[w * 16 + i] for i in range(10)]
        print('rep %d wave %d: layer 0 %6d | hidden layers %s | last %6d | pass %7d cycles  (per hidden group: %.1f)' % (
            rep, w, t[1] - t[0], ' '.join('%6d' % (t[i + 1] - t[i]) for i in range(1, 8)), t[9] - t[8], t[9] - t[0],
            (t[8] - t[1]) / (7 * 128.0)))
