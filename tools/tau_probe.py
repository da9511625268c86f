"""Distribution of the coarse (single-pass) evaluator's error |single pass - split| over 16.8 M points of the bounding
sphere for the SDF nets of configs 2-4, next to the bound the tracer uses (ops.calibrate_coarse_tau: 3 x the maximum over
65 536 points).  Usage: python tools/tau_probe.py"""
import sys, os, math
sys.path.insert(0, os.getcwd())
import torch
from nefii_amd import ops, synthetic as syn
from oracle import nets
for wl in ('cfg2', 'cfg3', 'cfg4'):
    mc, sd = syn.workload_state_dict(wl, seed=0)
    specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
    pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
    ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
    pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
    tau = ops.calibrate_coarse_tau(pm, 1.0)
    g = torch.Generator().manual_seed(123)
    errs = []
    for chunk in range(16):
        n = 1 << 20
        x = torch.randn(n, 3, generator=g)
        x = x / x.norm(dim=1, keepdim=True) * (torch.rand(n, 1, generator=g) ** (1.0 / 3.0)) * 1.02
        x = x.cuda()
        v = ops.sdf_eval(pm, x)
        e = (ops.sdf_eval(pm, x, coarse=True) - v).abs()
        errs.append(e)
        # near the surface (|sdf| < 0.02): where decisions are made
    e = torch.cat(errs)
    q = torch.quantile(e[:4000000].float(), torch.tensor([0.5, 0.99, 0.9999], device='cuda'))
    print('%s: tau(3x max of 65k) %.3e | 16.8M points: max %.3e  rms %.3e  median %.3e  p99 %.3e  p99.99 %.3e | max/rms %.1f  tau/max %.2f' % (
        wl, tau, e.max().item(), e.pow(2).mean().sqrt().item(), q[0].item(), q[1].item(), q[2].item(), e.max().item() / e.pow(2).mean().sqrt().item(), tau / e.max().item()))
