"""CPU study: how many of the min-SDF search's samples (ray_tracing.py:309-337: 100 random depths per ray that misses, argmin of
the SDF) need an evaluation at all if the SDF is L-Lipschitz along the ray?

Staged search simulated: the depths sorted; stage 1 evaluates every S-th of them (and both ends) in the single-pass arithmetic
(error < tau); an unevaluated sample s between evaluated neighbours a < s < b cannot be lower than
    lb(s) = max(v_a - L (t_s - t_a), v_b - L (t_b - t_s)) - tau
and is skipped when lb(s) > best + tau (best = the lowest stage-1 value: then s is not the argmin whatever its value is); stage
2 evaluates the rest; the refinement in split precision of the samples within 2 tau of the minimum is what it is today.
Prints evaluations per ray by (S, L), the share of rays where the staged search keeps the true argmin among its candidates (must
be all of them when L really bounds the slope), and the slopes the network really has along these rays.
Geometry: the benchmark's trained stand-in (configs 3-5) and the geometric-init sphere of configs 1-2.  Test infrastructure."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nefii_amd import synthetic as syn          # noqa: E402
from oracle import nets, renderer, tracer        # noqa: E402


def rays(workload, n_px, n_rays):
    w = syn.WORKLOADS[workload]
    inp, _ = syn.make_inputs(n_px, image_hw=w['image_hw'], focal=w['focal'], cam_pos=w['cam_pos'], num_rays=n_rays, seed=2)
    uv = inp['uv'].reshape(1, -1, 2)
    dirs, cam = renderer.camera_rays(uv, inp['pose'], inp['intrinsics'])
    dirs = dirs.reshape(-1, 3)
    return cam.reshape(1, 3).expand_as(dirs).contiguous(), dirs, inp['object_mask'].reshape(-1)


def main():
    torch.set_num_threads(8)
    for workload, n_px, n_rays, tau in (('cfg3', 384, 4, 1.2e-3), ('cfg2', 1536, 1, 2.4e-3)):
        mc, sd = syn.workload_state_dict(workload, seed=0)
        cfg = mc['implicit_network']
        sd64 = {k: v.double() for k, v in sd.items() if k.startswith('implicit_network')}
        sdf = lambda x: nets.sdf_forward(sd64, cfg, x.double())[:, 0].float()
        p = dict(tracer.DEFAULT_TRACER)
        p.update(syn.RAY_TRACER)
        o, d, om = rays(workload, n_px, n_rays)
        om = om.repeat_interleave(n_rays) if om.numel() * n_rays == o.shape[0] else om
        with torch.no_grad():
            res = tracer.trace(sdf, o, d, om.bool(), p, training=True)
            hit, samp, sph = res['hit'], res['sampler_mask'], res['sphere_hit']
            m = ((~hit & om.bool() & ~samp) | (~om.bool() & ~samp)) & sph
            t_io, _ = tracer.sphere_intersection(o, d, p['object_bounding_sphere'])
            # the search interval of sphere_trace's outputs: recompute as trace() does
            live_s, t_s, t_e, t_min, t_max = tracer.sphere_trace(sdf, o, d, sph, t_io, p, tracer.Counters())
            sel = hit & ~om.bool() & ~samp
            t_min = torch.where(sel, res['dists'], t_min)
            oo, dd, t0, t1 = o[m], d[m], t_min[m], t_max[m]
            steps = res['minsdf_steps'] if res['minsdf_steps'] is not None else torch.empty(p['n_steps']).uniform_(0, 1)
            order = torch.argsort(steps)
            ss = steps[order]
            ts = ss.unsqueeze(0) * (t1 - t0).unsqueeze(-1) + t0.unsqueeze(-1)            # [m, n] sorted along the ray
            n = ts.shape[1]
            pts = oo.unsqueeze(1) + ts.unsqueeze(-1) * dd.unsqueeze(1)
            v = sdf(pts.reshape(-1, 3)).reshape(-1, n)
        M = v.shape[0]
        slope = ((v[:, 1:] - v[:, :-1]).abs() / (ts[:, 1:] - ts[:, :-1]).clamp_min(1e-9))
        big = (ts[:, 1:] - ts[:, :-1]) > 1e-3
        print('%s: %d of %d rays search their min-SDF point; interval length mean %.2f; |slope| between neighbouring samples '
              '(gaps > 1e-3): max %.3f, 99.9 %% %.3f' % (workload, M, o.shape[0], float((t1 - t0).mean()), float(slope[big].max()),
                                                        float(slope[big].quantile(0.999))))
        g = torch.Generator().manual_seed(5)
        x = torch.randn(20000, 3, generator=g)
        x = x / x.norm(dim=1, keepdim=True) * torch.rand(20000, 1, generator=g) ** (1 / 3.0)
        gn = nets.sdf_gradient({k: v.float() for k, v in sd.items()}, cfg, x).norm(dim=1)
        print('   |grad sdf| on 20 000 points of the unit ball: max %.3f, 99.9 %% %.3f, mean %.3f' % (float(gn.max()), float(gn.quantile(0.999)), float(gn.mean())))
        rng = np.random.Generator(np.random.Philox(3))
        e = torch.from_numpy(rng.uniform(-tau / 3, tau / 3, size=tuple(v.shape)).astype(np.float32))
        vc = v + e                                   # single-pass values
        true_arg = v.argmin(1)
        for S in (3, 4, 5, 6, 8):
            s1 = torch.zeros(n, dtype=torch.bool)
            s1[::S] = True
            s1[-1] = True
            idx1 = torch.nonzero(s1).reshape(-1)
            for L in (1.0, 1.25, 1.5, 2.0):
                best = vc[:, idx1].min(1).values                       # [M]
                # neighbours of every sample among the stage-1 ones
                pos = torch.searchsorted(idx1, torch.arange(n), right=True) - 1
                a = idx1[pos.clamp(0, idx1.numel() - 1)]
                b = idx1[(pos + 1).clamp(0, idx1.numel() - 1)]
                lb = torch.maximum(vc[:, a] - L * (ts - ts[:, a]), vc[:, b] - L * (ts[:, b] - ts)) - tau
                keep = (lb <= best.unsqueeze(1) + tau) & ~s1.unsqueeze(0)
                evals = idx1.numel() + keep.sum(1).float()
                cand = s1.unsqueeze(0) | keep
                ok = cand[torch.arange(M), true_arg]
                print('   stride %d, L %.2f: %.1f single-pass evaluations per ray (of %d: x %.2f); true argmin kept on %d of %d rays' % (
                    S, L, float(evals.mean()), n, float(evals.mean()) / n, int(ok.sum()), M))


if __name__ == '__main__':
    main()
