"""CPU study for the tiered sphere tracing (VERDICT r4 next #1a): how many sphere-tracing evaluations can run on the
single-pass evaluator, by the oracle's tracer on config 3's camera with an error model of the coarse evaluator.

Tier rule simulated (what nefii_tracer.hip implements behind nefii_tracer_params.trace_tier):
  a step / back-off query goes to the single-pass evaluator when the step that led to it is > gate * tau (the first
  evaluation at the bounding sphere always does); its value v16 is ACCEPTED when |v16| > kappa * tau (then v > thr and the
  sign are certain: kappa >= 1 + thr / tau), else the query is repeated in split precision.
Prints the fraction of coarse queries, of repeated ones, and what the accepted coarse values do to the trace (hit flips,
|delta t| of converged rays, rays whose converged iteration changes).  Test / measurement infrastructure only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nefii_amd import synthetic          # noqa: E402
from oracle import nets, renderer, tracer        # noqa: E402


def small_sdf():
    small = dict(np.load(os.path.join(os.path.dirname(synthetic.__file__), 'assets', 'scene_bowl_sdf64.npz')))
    sd = {'implicit_network.' + k: torch.from_numpy(v.astype(np.float32)) for k, v in small.items()}
    cfg = dict(synthetic.CONF_MODEL['implicit_network'])
    cfg['dims'] = [64] * 8
    cfg['use_last_as_f'] = False
    return lambda x: nets.sdf_forward(sd, cfg, x)[:, 0]


def tiered_sphere_trace(sdf, o, d, hit, t_io, p, tau, kappa, gate, rng, err_rms):
    """oracle.tracer.sphere_trace with the tier; returns its outputs + statistics"""
    thr = p['sdf_threshold']
    st = dict(q=0, coarse=0, repeat=0)

    def ev(t, mask, step):
        """values of the masked queries; step: the step that led here (inf for the first evaluation)"""
        out = torch.zeros_like(t)
        if mask.any():
            v = sdf(tracer._pts(o[mask], t[mask], d[mask]))
            st['q'] += int(mask.sum())
            if tau > 0:
                go = step[mask] > gate * tau
                e = torch.from_numpy(rng.normal(0.0, err_rms, size=v.shape).astype(np.float32)).clamp(-tau / 3, tau / 3)
                v16 = v + e
                acc = go & (v16.abs() > kappa * tau)
                st['coarse'] += int(go.sum())
                st['repeat'] += int((go & ~acc).sum())
                v = torch.where(acc, v16, v)
            out[mask] = v
        return out

    inf = torch.full_like(t_io[:, 0], float('inf'))
    t_s = torch.where(hit, t_io[:, 0], torch.zeros(()))
    t_e = torch.where(hit, t_io[:, 1], torch.zeros(()))
    live_s, live_e = hit.clone(), hit.clone()
    nxt_s, nxt_e = ev(t_s, live_s, inf), ev(t_e, live_e, inf)
    it = 0
    conv_it = torch.full(t_s.shape, -1, dtype=torch.long)
    while True:
        cur_s = torch.where(live_s, nxt_s, torch.zeros(()))
        cur_s = torch.where(cur_s <= thr, torch.zeros(()), cur_s)
        cur_e = torch.where(live_e, nxt_e, torch.zeros(()))
        cur_e = torch.where(cur_e <= thr, torch.zeros(()), cur_e)
        was = live_s.clone()
        live_s = live_s & (cur_s > thr)
        live_e = live_e & (cur_e > thr)
        conv_it = torch.where(was & ~live_s & (conv_it < 0), torch.full_like(conv_it, it), conv_it)
        if it == p['sphere_tracing_iters'] or not (live_s.any() or live_e.any()):
            break
        it += 1
        t_s = t_s + cur_s
        t_e = t_e - cur_e
        nxt_s, nxt_e = ev(t_s, live_s, cur_s), ev(t_e, live_e, cur_e)
        bad_s, bad_e = nxt_s < 0, nxt_e < 0
        k = 0
        while (bad_s.any() or bad_e.any()) and k < p['line_step_iters']:
            back = (1 - p['line_search_step']) / (2 ** k)
            t_s = torch.where(bad_s, t_s - back * cur_s, t_s)
            t_e = torch.where(bad_e, t_e + back * cur_e, t_e)
            if bad_s.any():
                nxt_s = torch.where(bad_s, ev(t_s, bad_s, back * cur_s), nxt_s)
            if bad_e.any():
                nxt_e = torch.where(bad_e, ev(t_e, bad_e, back * cur_e), nxt_e)
            bad_s, bad_e = nxt_s < 0, nxt_e < 0
            k += 1
        live_s = live_s & (t_s < t_e)
        live_e = live_e & (t_s < t_e)
    return live_s, t_s, t_e, conv_it, st


def rays_primary(n_px, n_rays):
    w = synthetic.WORKLOADS['cfg3']
    inp, _ = synthetic.make_inputs(n_px, image_hw=w['image_hw'], focal=w['focal'], cam_pos=w['cam_pos'], num_rays=n_rays, seed=2)
    uv = inp['uv'].reshape(1, -1, 2)
    dirs, cam = renderer.camera_rays(uv, inp['pose'], inp['intrinsics'])
    dirs = dirs.reshape(-1, 3)
    return cam.reshape(1, 3).expand_as(dirs).contiguous(), dirs


def main():
    torch.set_num_threads(8)
    sdf = small_sdf()
    p = dict(tracer.DEFAULT_TRACER)
    p.update(synthetic.RAY_TRACER)
    tau = 2.4e-3
    o, d = rays_primary(256, 16)
    with torch.no_grad():
        sets = [('primary', o, d)]
        # secondary-like rays: from the primary hit points into a random hemisphere about the normal
        res = tracer.trace(sdf, o, d, torch.ones(o.shape[0], dtype=torch.bool), p, training=False)
        hp = res['points'][res['hit']]
        g = torch.Generator().manual_seed(3)
        n = torch.nn.functional.normalize(nets.sdf_gradient({'implicit_network.' + k: torch.from_numpy(v.astype(np.float32)) for k, v in dict(np.load(os.path.join(os.path.dirname(synthetic.__file__), 'assets', 'scene_bowl_sdf64.npz'))).items()},
                                                             dict(synthetic.CONF_MODEL['implicit_network'], dims=[64] * 8, use_last_as_f=False), hp), dim=1)
        w = torch.nn.functional.normalize(torch.randn(hp.shape[0], 3, generator=g), dim=1)
        w = torch.where((w * n).sum(1, keepdim=True) < 0, -w, w)
        sets.append(('secondary', hp, w))
        for name, oo, dd in sets:
            t_io, sph = tracer.sphere_intersection(oo, dd, p['object_bounding_sphere'])
            ref = tiered_sphere_trace(sdf, oo, dd, sph, t_io, p, 0.0, 0, 0, None, 0)
            print('%s: %d rays, %d sphere-tracing evaluations (%.1f per ray)' % (name, oo.shape[0], ref[4]['q'], ref[4]['q'] / oo.shape[0]))
            for kappa, gate in ((1.05, 0.0), (2.0, 0.0), (2.0, 4.0), (2.0, 8.0), (3.0, 8.0), (2.0, 16.0)):
                rng = np.random.Generator(np.random.Philox(7))
                live, t_s, t_e, cit, st = tiered_sphere_trace(sdf, oo, dd, sph, t_io, p, tau, kappa, gate, rng, 1.7e-4)
                hit0, hit1 = ref[1] < ref[2], t_s < t_e
                conv = (~ref[0]) & (~live) & hit0 & hit1
                dt = (t_s - ref[1]).abs()[conv]
                print('  kappa %.2f gate %4.1f: queries %d (%.3f of ref), coarse %.3f, repeated %.3f of coarse -> issued-MFMA %.3f of ref | '
                      'hit flips %d, sampler-set flips %d, converged-iteration changes %d of %d, |dt| converged max %.2e mean %.2e' % (
                          kappa, gate, st['q'], st['q'] / ref[4]['q'], st['coarse'] / st['q'], st['repeat'] / max(1, st['coarse']),
                          ((st['q'] - st['coarse'] + st['repeat']) * 3 + st['coarse']) / (3.0 * ref[4]['q']),
                          int((hit0 != hit1).sum()), int((ref[0] != live).sum()), int((cit != ref[3])[conv].sum()), int(conv.sum()),
                          float(dt.max()) if dt.numel() else 0.0, float(dt.mean()) if dt.numel() else 0.0))


if __name__ == '__main__':
    main()
