"""Layer-by-layer check of the register-resident coarse evaluator against an fp64 forward (debug build:
tools/ab_build.sh dbg -DNEFII_C_DEBUG; NEFII_LIB_PATH=build_ab/libnefii_dbg.so python tools/coarse_x_debug.py)."""
import ctypes
import math
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
n = 256
x = (torch.randn(n, 3, generator=torch.Generator().manual_seed(1)) * 0.4)
# fp64 forward with every layer's activations (implicit_differentiable_renderer.py:85-108: PE, skip at layer 4)
xd = x.double()
pe = [xd]
for k in range(6):
    pe += [torch.sin(xd * 2.0 ** k), torch.cos(xd * 2.0 ** k)]
pe = torch.cat(pe, -1)
skip = [l for l, s in enumerate(specs) if s.e_len > 0 and s.x_len > 0]
h = pe
acts, inputs = [], []
for l, (w, b) in enumerate(zip(ws, bs)):
    if l in skip:
        h = torch.cat([h, pe], -1) / math.sqrt(2)
    inputs.append(h)
    z = h @ w.double().t() + b.double()
    if l < len(ws) - 1:
        h = torch.nn.functional.softplus(z, beta=100)
        acts.append(h)
lib = _lib.lib()
lib.nefii_debug_c_select.argtypes = [ctypes.c_int]
xg = x.cuda()
for l in range(int(os.environ.get('DBG_LAYERS', len(acts)))):
    width = acts[l].shape[1]
    for k in (list(range(0, 40)) + [100, 255, 256, width - 1] if l == 0 else [0, 1, 4, 8, 12, 15, 16, 31, 100, 255, 256, width - 1]):
        lib.nefii_debug_c_select(((l + 1) << 10) | k)
        got = ops.sdf_eval(pm, xg, coarse=True).cpu().double() / 16.0
        ref = acts[l][:, k]
        print('layer %d col %3d: max|got-ref| %.3e   (ref rms %.3e, got rms %.3e)' % (l, k, (got - ref).abs().max(), ref.pow(2).mean().sqrt(), got.pow(2).mean().sqrt()))
    if l >= 1:
        for k in [0, 100, 464, 471, 472, 473, 474, 479, 480, 495, 496, 511]:
            lib.nefii_debug_c_select(((l + 17) << 10) | k)
            got = ops.sdf_eval(pm, xg, coarse=True).cpu().double() / 16.0
            ref = inputs[l][:, k] * (math.sqrt(2) if l in skip else 1.0)
            print('  input of layer %d col %3d: max|got-ref| %.3e   (ref rms %.3e)' % (l, k, (got - ref).abs().max(), ref.pow(2).mean().sqrt()))
lib.nefii_debug_c_select(0)
