"""How much does the local search of ops.calibrate_lipschitz raise the largest |grad sdf| found by the random sample alone?
(GPU; per geometry: random 65 536 points, + 1 ... 6 rounds of search, and a 4 M-point random sample for comparison.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nefii_amd import ops, synthetic as syn
from oracle import nets

for name, scene, bumpy in (('conf', 'bowl_trained', 0.0), ('conf', 'frame_trained', 0.0), ('neus', 'bowl_trained', 0.0),
                           ('conf', 'bowl_dense', 0.0), ('physg', None, 0.004), ('physg', None, 0.0)):
    mc = syn.model_conf(name)
    sd = syn.make_state_dict(mc, seed=0, bumpy=bumpy, scene=scene)
    specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
    pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda')
    ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
    pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
    gf = lambda x: ops.sdf_value_grad(pm, x)[2]
    row = ['%.4f' % ops.calibrate_lipschitz(gf, 'cuda', safety=1.0, search_rounds=r) for r in (0, 1, 2, 4, 6)]
    g = torch.Generator().manual_seed(99)
    big = 0.0
    for _ in range(16):
        x = torch.randn(1 << 18, 3, generator=g)
        x = (x / x.norm(dim=1, keepdim=True) * (torch.rand(1 << 18, 1, generator=g) ** (1 / 3.0)) * 1.02).cuda()
        big = max(big, gf(x).norm(dim=1).max().item())
    print('%-6s %-14s bumpy %.3f: largest |grad sdf| by search rounds 0 / 1 / 2 / 4 / 6: %s; 4.2 M random points: %.4f' % (
        name, scene, bumpy, ' / '.join(row), big))
