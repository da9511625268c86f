"""nefii_mlp_forward_f16 (split-precision forward of the radiance and material MLPs, with / without the stash) and
nefii_mlp_backward_f16 (one fp16 pass) - time per call at n points.  NEFII_MLP_STREAM=0 selects the 32-row kernel.  Usage: python tools/mlp_microbench.py [n ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import ops, synthetic as syn
from oracle import nets

name = os.environ.get('MODEL', 'conf')
mc = syn.model_conf(name)
sd = syn.make_state_dict(mc, seed=1)
F = mc['feature_vector_size']
dev = 'cuda'
g = torch.Generator().manual_seed(5)


def packed(kind):
    if kind == 'radiance':
        specs, enc, head = ops.radiance_specs(mc['rendering_network'], F)
        pm = ops.PackedMLP(specs, ops.ACT_RELU, head, enc, F, dev, half='f16x3')
        ws, bs = zip(*[nets.linear_params(sd, 'rendering_network.lin%d' % l) for l in range(len(specs))])
    else:
        mcfg = mc['envmap_material_network']
        specs, enc = ops.material_specs(mcfg, F, 4 if mcfg.get('roughness_mlp') else 3)
        pm = ops.PackedMLP(specs, ops.ACT_ELU, ops.HEAD_SIGMOID, enc, F, dev, half='f16x3')
        lp = 'envmap_material_network.diffuse_albedo_layers'
        ws = [sd['%s.%d.weight' % (lp, 2 * l)] for l in range(len(specs))]
        bs = [sd['%s.%d.bias' % (lp, 2 * l)] for l in range(len(specs))]
    pm.pack([w.to(dev) for w in ws], [b.to(dev) for b in bs])
    return pm


for kind in ('radiance', 'material'):
    pm = packed(kind)
    for n in [int(a) for a in sys.argv[1:]] or [4096, 139264]:
        x = (torch.randn(n, 3, generator=g) * 0.4).to(dev)
        v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        nr = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        feat = (torch.randn(n, F, generator=g) * 0.3).to(dev)
        args = (pm, x, v, nr, feat) if kind == 'radiance' else (pm, x, None, None, feat)
        ms = {}
        for want in (True, False):          # training mode (activations stashed for the backward pass) / evaluation
            for _ in range(3):
                out, hid, stash = ops.mlp_forward(*args, want_stash=want)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                out, hid, stash = ops.mlp_forward(*args, want_stash=want)
            e1.record()
            torch.cuda.synchronize()
            ms[want] = e0.elapsed_time(e1) / 10
        out, hid, stash = ops.mlp_forward(*args, want_stash=True)
        d_out = torch.randn_like(out) * 1e-6
        gs = ops.mlp_grad_scale(d_out)
        for _ in range(3):
            dz = ops.mlp_backward(pm, d_out, stash, gs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dz = ops.mlp_backward(pm, d_out, stash, gs)
        e1.record()
        torch.cuda.synchronize()
        print('%-9s %s  n %7d  forward %.3f ms per call with stash, %.3f without; backward %.3f   streamed %s   finite %s' % (
            kind, name, n, ms[True], ms[False], e0.elapsed_time(e1) / 10, pm.mlp_stream,
            # (of the head's dz row only its n_pad <= 32 columns are written)
            bool(torch.isfinite(out).all() and torch.isfinite(dz[:-1]).all() and torch.isfinite(dz[-1][:, :out.shape[1]]).all())))
