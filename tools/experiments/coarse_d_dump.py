"""Dumps the single-pass evaluator's values on a fixed point set (NEFII_COARSE_D selects the tile form): bit-identity A/B.
Usage: NEFII_COARSE_D=0|1 python tools/experiments/coarse_d_dump.py out.npy [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from nefii_amd import ops, synthetic as syn
from oracle import nets

mc = syn.model_conf('physg')
sd = syn.make_state_dict(mc, seed=0, bumpy=0.0, scene=os.environ.get('SCENE', 'bowl_trained'))
specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200003
x = (torch.randn(n, 3, generator=torch.Generator().manual_seed(5)) * 0.5).cuda()
out = ops.sdf_eval(pm, x, coarse=True)
ref = ops.sdf_eval(pm, x, coarse=False)
torch.cuda.synchronize()
print('NEFII_COARSE_D=%s  n %d  max |single pass - split| %.3e  finite %s' % (
    os.environ.get('NEFII_COARSE_D', '0'), n, (out - ref).abs().max().item(), bool(torch.isfinite(out).all())))
np.save(sys.argv[1], out.cpu().numpy())
