"""CPU study for DESIGN section 7 (2): what would the split evaluator's two correction products on block-scaled fp8 do to the
TRACER's decisions?  The oracle's tracer (both-ends sphere tracing, bracket search, bisection) runs on config 3's primary rays
and on secondary-like rays with the SDF evaluated in an emulated arithmetic (tools/experiments/arith_emulation.py):
    split    x_h w_h + x_h w_l + x_l w_h on fp16 hi/lo pairs (today's split evaluator; control)
    fp8corr  x_h w_h + x_h q(w_l) + q(x_l) w_h, q = e4m3 with one power-of-two scale per 32 K
against the same tracer on the fp32 network: hit-mask flips, depth differences of rays that hit both ways, rays that change
path (converged by sphere tracing on one side, through the bracket search on the other).  Eval mode (no min-SDF search).
Usage: python tools/experiments/fp8corr_trace_stats.py [n_pixels=192]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import arith_emulation as ae                      # noqa: E402
from nefii_amd import synthetic as syn          # noqa: E402
from oracle import nets, renderer, tracer        # noqa: E402


def main():
    torch.set_num_threads(8)
    n_px = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    mc, sd = syn.workload_state_dict('cfg3', seed=0)
    cfg = mc['implicit_network']
    p = dict(tracer.DEFAULT_TRACER)
    p.update(syn.RAY_TRACER)
    w = syn.WORKLOADS['cfg3']
    inp, _ = syn.make_inputs(n_px, image_hw=w['image_hw'], focal=w['focal'], cam_pos=w['cam_pos'], num_rays=4, seed=2)
    dirs, cam = renderer.camera_rays(inp['uv'].reshape(1, -1, 2), inp['pose'], inp['intrinsics'])
    d = dirs.reshape(-1, 3)
    o = cam.reshape(1, 3).expand_as(d).contiguous()
    om = torch.ones(o.shape[0], dtype=torch.bool)
    f32 = lambda x: nets.sdf_forward(sd, cfg, x)[:, 0]
    with torch.no_grad():
        base = tracer.trace(f32, o, d, om, p, training=False)
        hp = base['points'][base['hit']]
        g = torch.Generator().manual_seed(3)
        nrm = torch.nn.functional.normalize(nets.sdf_gradient(sd, cfg, hp), dim=1)
        wdir = torch.nn.functional.normalize(torch.randn(hp.shape[0], 3, generator=g), dim=1)
        wdir = torch.where((wdir * nrm).sum(1, keepdim=True) < 0, -wdir, wdir)
        sets = [('primary', o, d, om, base), ('secondary', hp, wdir, torch.ones(hp.shape[0], dtype=torch.bool), None)]
        for name, oo, dd, mm, ref in sets:
            if ref is None:
                ref = tracer.trace(f32, oo, dd, mm, p, training=False)
            print('%s: %d rays, hit fraction %.3f, %.3f through the bracket search' % (name, oo.shape[0], float(ref['hit'].float().mean()),
                                                                                     float(ref['sampler_mask'].float().mean())))
            for mode in ('split', 'fp8corr'):
                emu = lambda x, mode=mode: ae.sdf_forward_emu(sd, cfg, x, mode)[:, 0]
                got = tracer.trace(emu, oo, dd, mm, p, training=False)
                flips = int((got['hit'] != ref['hit']).sum())
                both = got['hit'] & ref['hit']
                dt = (got['dists'] - ref['dists']).abs()[both]
                path = int((got['sampler_mask'] != ref['sampler_mask']).sum())
                print('   %-8s hit-mask flips %d | both hit %d: |d depth| max %.2e mean %.2e, > 1e-5: %.4f, > 1e-4: %.4f | rays changing path %d' % (
                    mode, flips, int(both.sum()), float(dt.max()), float(dt.mean()), float((dt > 1e-5).float().mean()),
                    float((dt > 1e-4).float().mean()), path))


if __name__ == '__main__':
    main()
