"""CPU study: the bracket search (ray_tracing.py:195-257: 100 evenly spaced samples between the two sphere-tracing fronts of a ray
that did not converge, first negative sample, argmin fallback) with a Lipschitz-staged evaluation against today's quarter-row
windows, on SECONDARY rays of the trained stand-in (hemisphere directions from the primary hit points) - where the search is
5.96 M of config 3's 18.4 M single-pass evaluations per step (profiles/r05/rounds_cfg3_secondary.txt).

  windows  today: quarter rows of 25 consecutive samples until one holds a surely negative sample (< -tau)
  staged   25 samples spread over the row (both ends) first; a sample s between evaluated neighbours is skipped when its lower
           bound lb(s) = max(c_a - L dt_a, c_b - L dt_b) - tau proves it positive AND not the argmin: lb > max(0, best + tau)
           - or, once a surely negative sample j1 is known (the ray lies inside the mask: no argmin), when lb > 0; samples behind
           j1 are not needed at all
Prints samples evaluated per search by both, and checks that the staged search keeps the first negative sample / the argmin."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nefii_amd import synthetic as syn          # noqa: E402
from oracle import nets, renderer, tracer        # noqa: E402


def main():
    torch.set_num_threads(8)
    mc, sd = syn.workload_state_dict('cfg3', seed=0)
    cfg = mc['implicit_network']
    sd64 = {k: v.double() for k, v in sd.items() if k.startswith('implicit_network')}
    sdf = lambda x: nets.sdf_forward(sd64, cfg, x.double())[:, 0].float()
    p = dict(tracer.DEFAULT_TRACER)
    p.update(syn.RAY_TRACER)
    tau, n = 1.2e-3, p['n_steps']
    w = syn.WORKLOADS['cfg3']
    inp, _ = syn.make_inputs(512, image_hw=w['image_hw'], focal=w['focal'], cam_pos=w['cam_pos'], num_rays=4, seed=2)
    dirs, cam = renderer.camera_rays(inp['uv'].reshape(1, -1, 2), inp['pose'], inp['intrinsics'])
    d = dirs.reshape(-1, 3)
    o = cam.reshape(1, 3).expand_as(d).contiguous()
    with torch.no_grad():
        res = tracer.trace(sdf, o, d, torch.ones(o.shape[0], dtype=torch.bool), p, training=False)
        hp = res['points'][res['hit']]
        g = torch.Generator().manual_seed(3)
        nrm = torch.nn.functional.normalize(nets.sdf_gradient({k: v.float() for k, v in sd.items()}, cfg, hp), dim=1)
        wdir = torch.nn.functional.normalize(torch.randn(hp.shape[0] * 3, 3, generator=g), dim=1)
        hp3, n3 = hp.repeat_interleave(3, 0), nrm.repeat_interleave(3, 0)
        wdir = torch.where((wdir * n3).sum(1, keepdim=True) < 0, -wdir, wdir)
        t_io, sph = tracer.sphere_intersection(hp3, wdir, p['object_bounding_sphere'])
        t_io[:, 0] = 0.01                                           # secondary rays start 0.01 off the surface (as the renderer sends them)
        live_s, t_s, t_e, _, _ = tracer.sphere_trace(sdf, hp3, wdir, sph, t_io, p, tracer.Counters())
        oo, dd, a, b = hp3[live_s], wdir[live_s], t_s[live_s], t_e[live_s]
        lin = torch.linspace(0, 1, n)
        ts = a.unsqueeze(1) + lin.view(1, -1) * (b - a).unsqueeze(1)
        v = sdf((oo.unsqueeze(1) + ts.unsqueeze(-1) * dd.unsqueeze(1)).reshape(-1, 3)).reshape(-1, n)
    M = v.shape[0]
    print('%d secondary rays, %d in the bracket search (%.2f); interval length mean %.3f' % (hp3.shape[0], M, M / hp3.shape[0], float((b - a).mean())))
    rng = np.random.Generator(np.random.Philox(3))
    c = v + torch.from_numpy(rng.uniform(-tau / 3, tau / 3, size=tuple(v.shape)).astype(np.float32))
    neg = v < 0
    first_neg = torch.where(neg.any(1), neg.float().argmax(1), torch.full((M,), -1))
    print('   rays with a negative sample: %.3f; first negative sample index: median %d' % (float((first_neg >= 0).float().mean()),
                                                                                          int(first_neg[first_neg >= 0].median())))
    # today: windows
    sure = c < -tau
    win = torch.zeros(M)
    for r in range(M):
        k = 0
        for wdw in range(4):
            k = wdw + 1
            if sure[r, :25 * (wdw + 1)].any() and not (c[r, 0] < tau):
                break
        win[r] = 25 * k
    print('   today (quarter rows): %.1f single-pass samples per search' % float(win.mean()))
    idx1 = torch.tensor(sorted(set(int(round(j * (n - 1) / 24)) for j in range(25))))
    s1 = torch.zeros(n, dtype=torch.bool)
    s1[idx1] = True
    dt = (b - a) / (n - 1)
    for L in (1.25, 1.5, 2.0):
        best = c[:, idx1].min(1).values
        pos = torch.searchsorted(idx1, torch.arange(n), right=True) - 1
        ia = idx1[pos.clamp(0, idx1.numel() - 1)]
        ib = idx1[(pos + 1).clamp(0, idx1.numel() - 1)]
        k = torch.arange(n).view(1, -1)
        lb = torch.maximum(c[:, ia] - L * dt.unsqueeze(1) * (k - ia.view(1, -1)), c[:, ib] - L * dt.unsqueeze(1) * (ib.view(1, -1) - k)) - tau
        sure1 = (c < -tau) & s1.view(1, -1)
        j1 = torch.where(sure1.any(1), sure1.float().argmax(1), torch.full((M,), n))          # first surely negative stage-1 sample
        thr = torch.where(j1 < n, torch.zeros(M), torch.clamp(best + tau, min=0.0)).unsqueeze(1)
        keep = (lb <= thr + 1e-6) & ~s1.view(1, -1) & (k < j1.view(-1, 1))
        evals = 25 + keep.sum(1).float()
        cand = s1.view(1, -1) | keep
        # checks: the first negative sample is among the candidates; for rays without one the argmin is
        ok_neg = torch.ones(M, dtype=torch.bool)
        has = first_neg >= 0
        ok_neg[has] = cand[has, first_neg[has]]
        amin = v.argmin(1)
        ok_min = cand[torch.arange(M), amin] | has
        print('   staged, L %.2f: %.1f samples per search (x %.2f of today); first negative sample kept %d / %d, argmin of rays without one kept %d / %d' % (
            L, float(evals.mean()), float(evals.mean() / win.mean()), int(ok_neg[has].sum()), int(has.sum()), int(ok_min[~has].sum()), int((~has).sum())))


if __name__ == '__main__':
    main()
