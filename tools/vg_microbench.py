"""nefii_sdf_value_grad (value + features + normals of surface points) on the physg SDF net: time per call and error
against the fp64 oracle.  NEFII_VG_STREAM=0 selects the generic 32-row kernel.  Usage: python tools/vg_microbench.py [n ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import ops, synthetic as syn
from oracle import nets

mc = syn.model_conf(os.environ.get('MODEL', 'physg'))
sd = syn.make_state_dict(mc, seed=0, bumpy=0.004)
specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
g = torch.Generator().manual_seed(1)
sd64 = {k: v.double() for k, v in sd.items()}
for n in [int(a) for a in sys.argv[1:]] or [4096, 139264]:
    x = (torch.randn(n, 3, generator=g) * 0.4).cuda()
    for _ in range(3):
        out, feat, grad = ops.sdf_value_grad(pm, x, want_feat=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        out, feat, grad = ops.sdf_value_grad(pm, x, want_feat=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    m = min(n, 1000) - 7          # a ragged prefix ending inside a tile
    xr = x[:m].cpu().double().requires_grad_(True)
    y = nets.sdf_forward(sd64, mc['implicit_network'], xr)
    gr, = torch.autograd.grad(y[:, 0].sum(), xr)
    if y.shape[1] == 1 + feat.shape[1]:          # use_last_as_f nets: the feature columns are the last hidden activation
        ferr = (feat[:m].cpu().double() - y[:, 1:].detach()).abs().max().item()
    elif out.shape[1] > 1:                       # the last layer carries the feature columns itself
        ferr = (out[:m, 1:].cpu().double() - y[:, 1:].detach()).abs().max().item()
    else:
        ferr = float('nan')
    print('n %7d  %.3f ms per call   max|sdf err| %.2e   max|grad err| %.2e   max|feature err| %.2e   all rows finite %s' % (
        n, ms, (out[:m, 0].cpu().double() - y[:, 0].detach()).abs().max().item(),
        (grad[:m].cpu().double() - gr).abs().max().item(), ferr,
        bool(torch.isfinite(grad).all() and torch.isfinite(feat).all())))
