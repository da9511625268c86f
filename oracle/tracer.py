"""Oracle: bounding-sphere intersection + SDF ray tracer, per-ray semantics.

TEST INFRASTRUCTURE (see oracle/__init__.py).  fp32 PyTorch-CPU restatement of
  * rend_util.get_sphere_intersection   code/utils/rend_util.py:200-221
  * RayTracing.forward                  code/model/ray_tracing.py:29-101
  * sphere_tracing                      code/model/ray_tracing.py:104-193
  * ray_sampler + rootfind              code/model/ray_tracing.py:195-280
  * minimal_sdf_points                  code/model/ray_tracing.py:309-337
Rays carry their own origin (primary rays share the camera centre, secondary
rays start at surface points: path_tracing_render.py:1351-1354).

Every arithmetic step keeps the reference's operation order (mul then add for
``o + t*d``; float32 throughout) so the restatement is bit-comparable with the
reference wherever the reference is batch-independent.
"""
import torch

DEFAULT_TRACER = dict(object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5,
                      line_step_iters=1, sphere_tracing_iters=10, n_steps=100, n_rootfind_steps=8)


def sphere_intersection(origins, dirs, r=1.0):
    """t_near/t_far of the |x|=r sphere along unit rays; clamp_min(0.01) (rend_util.py:200-221).

    origins, dirs: [N,3].  Returns t [N,2] and hit mask [N]."""
    b = (dirs * origins).sum(-1)
    under = b ** 2 - (origins.norm(2, 1) ** 2 - r ** 2)
    hit = under > 0
    t = torch.zeros(origins.shape[0], 2)
    root = torch.sqrt(under[hit])
    t[hit, 0] = -root
    t[hit, 1] = root
    t[hit] -= b[hit].unsqueeze(-1)
    return t.clamp_min(0.01), hit


class Counters(dict):
    def add(self, key, n):
        self[key] = self.get(key, 0) + int(n)


def _pts(o, t, d):
    return o + t.unsqueeze(-1) * d


def _masked_sdf(sdf, o, t, d, mask, cnt, key):
    out = torch.zeros_like(t)
    if mask.any():
        out[mask] = sdf(_pts(o[mask], t[mask], d[mask]))
        cnt.add(key, mask.sum())
    return out


def sphere_trace(sdf, o, d, hit, t_io, p, cnt):
    """Both-ends sphere tracing with back-off line search (ray_tracing.py:104-193)."""
    thr = p['sdf_threshold']
    t_s = torch.where(hit, t_io[:, 0], torch.zeros(()))
    t_e = torch.where(hit, t_io[:, 1], torch.zeros(()))
    live_s = hit.clone()
    live_e = hit.clone()
    t_min = t_s.clone()
    t_max = t_e.clone()
    nxt_s = _masked_sdf(sdf, o, t_s, d, live_s, cnt, 'sphere_trace')
    nxt_e = _masked_sdf(sdf, o, t_e, d, live_e, cnt, 'sphere_trace')
    it = 0
    while True:
        cur_s = torch.where(live_s, nxt_s, torch.zeros(()))
        cur_s = torch.where(cur_s <= thr, torch.zeros(()), cur_s)
        cur_e = torch.where(live_e, nxt_e, torch.zeros(()))
        cur_e = torch.where(cur_e <= thr, torch.zeros(()), cur_e)
        live_s = live_s & (cur_s > thr)
        live_e = live_e & (cur_e > thr)
        if it == p['sphere_tracing_iters'] or not (live_s.any() or live_e.any()):
            break
        it += 1
        t_s = t_s + cur_s
        t_e = t_e - cur_e
        nxt_s = _masked_sdf(sdf, o, t_s, d, live_s, cnt, 'sphere_trace')
        nxt_e = _masked_sdf(sdf, o, t_e, d, live_e, cnt, 'sphere_trace')
        bad_s = nxt_s < 0
        bad_e = nxt_e < 0
        k = 0
        while (bad_s.any() or bad_e.any()) and k < p['line_step_iters']:
            back = (1 - p['line_search_step']) / (2 ** k)
            t_s = torch.where(bad_s, t_s - back * cur_s, t_s)
            t_e = torch.where(bad_e, t_e + back * cur_e, t_e)
            if bad_s.any():
                nxt_s[bad_s] = sdf(_pts(o[bad_s], t_s[bad_s], d[bad_s]))
                cnt.add('sphere_trace', bad_s.sum())
            if bad_e.any():
                nxt_e[bad_e] = sdf(_pts(o[bad_e], t_e[bad_e], d[bad_e]))
                cnt.add('sphere_trace', bad_e.sum())
            bad_s = nxt_s < 0
            bad_e = nxt_e < 0
            k += 1
        live_s = live_s & (t_s < t_e)
        live_e = live_e & (t_s < t_e)
    return live_s, t_s, t_e, t_min, t_max


def first_crossing(vals):
    """Index of the first negative sample; first exact zero if none; last sample otherwise.

    Restates argmin(sign(sdf) * [n..1]) (ray_tracing.py:218-219)."""
    n = vals.shape[1]
    key = torch.sign(vals) * torch.arange(n, 0, -1, dtype=vals.dtype).reshape(1, n)
    return torch.argmin(key, -1)


def bisect(sdf, o, d, lo, hi, f_lo, f_hi, p, cnt):
    """Per-ray bisection of [lo, hi] (ray_tracing.py:259-280), stopping per ray."""
    work = (f_lo > 0) & (f_hi < 0) & (hi > lo)
    mid = (lo + hi) / 2.
    i = 0
    while work.any() and i < p['n_rootfind_steps']:
        f_mid = torch.zeros_like(mid)
        f_mid[work] = sdf(_pts(o[work], mid[work], d[work]))
        cnt.add('bisect', work.sum())
        up = work & (f_mid > 0)
        dn = work & ~(f_mid > 0)
        lo = torch.where(up, mid, lo)
        hi = torch.where(dn, mid, hi)
        mid = torch.where(work, (lo + hi) / 2., mid)
        work = work & ((hi - lo) > 1e-6)
        i += 1
    return mid


def sample_and_root(sdf, o, d, t_s, t_e, object_mask, training, p, cnt, lin=None):
    """Uniform sampler + bracket + bisection for not-converged rays (ray_tracing.py:195-257).

    Returns (dist, net_hit) for the given rays."""
    n = p['n_steps']
    if lin is None:
        lin = torch.linspace(0, 1, steps=n)
    ts = t_s.unsqueeze(-1) + lin.view(1, -1) * (t_e - t_s).unsqueeze(-1)          # [m, n]
    pts = o.unsqueeze(1) + ts.unsqueeze(-1) * d.unsqueeze(1)                      # [m, n, 3]
    vals = sdf(pts.reshape(-1, 3)).reshape(-1, n)
    cnt.add('sampler', vals.numel())
    rows = torch.arange(vals.shape[0])
    ind = first_crossing(vals)
    dist = ts[rows, ind]
    net_hit = vals[rows, ind] < 0
    p_out = ~(object_mask & net_hit)
    if p_out.any():
        amin = torch.argmin(vals, -1)
        dist = torch.where(p_out, ts[rows, amin], dist)
    root = (net_hit & object_mask) if training else net_hit
    if root.any():
        r = rows[root]
        hi = ts[r, ind[root]]
        f_hi = vals[r, ind[root]]
        lo = ts[r, ind[root] - 1]            # index -1 wraps to the last sample (quirk :245-246)
        f_lo = vals[r, ind[root] - 1]
        z = bisect(sdf, o[root], d[root], lo.clone(), hi.clone(), f_lo, f_hi, p, cnt)
        dist = dist.clone()
        dist[root] = z
    return dist, net_hit


def min_sdf_search(sdf, o, d, t_min, t_max, steps, cnt):
    """n random depths shared by all rays, argmin SDF (ray_tracing.py:309-337)."""
    ts = steps.unsqueeze(0) * (t_max - t_min).unsqueeze(-1) + t_min.unsqueeze(-1)
    pts = o.unsqueeze(1) + ts.unsqueeze(-1) * d.unsqueeze(1)
    vals = sdf(pts.reshape(-1, 3)).reshape(-1, steps.shape[0])
    cnt.add('min_sdf', vals.numel())
    idx = vals.argmin(-1)
    return ts[torch.arange(ts.shape[0]), idx]


def trace(sdf, origins, dirs, object_mask, params, training, minsdf_steps=None, counters=None):
    """RayTracing.forward for rays with per-ray origins.

    sdf: callable [n,3] -> [n].  object_mask: bool [N].  ``minsdf_steps``: the n_steps
    uniforms of minimal_sdf_points (drawn from torch's global RNG when None, exactly
    where the reference draws them: only if the masked set is non-empty).
    Returns dict(points, hit, dists, sphere_hit, sampler_mask, counters)."""
    p = dict(DEFAULT_TRACER)
    p.update(params or {})
    cnt = counters if counters is not None else Counters()
    with torch.no_grad():
        t_io, sph = sphere_intersection(origins, dirs, p['object_bounding_sphere'])
        live_s, t_s, t_e, t_min, t_max = sphere_trace(sdf, origins, dirs, sph, t_io, p, cnt)
        hit = t_s < t_e
        dist = t_s.clone()
        samp = live_s
        if samp.any():
            sd_, sh_ = sample_and_root(sdf, origins[samp], dirs[samp], t_s[samp], t_e[samp],
                                       object_mask[samp], training, p, cnt)
            dist[samp] = sd_
            hit = hit.clone()
            hit[samp] = sh_
        if training:
            in_m = ~hit & object_mask & ~samp
            out_m = ~object_mask & ~samp
            left = (in_m | out_m) & ~sph
            if left.any():           # closest point of the ray to the origin (:82-87)
                dist[left] = -(dirs[left] * origins[left]).sum(-1)
            m = (in_m | out_m) & sph
            if m.any():
                sel = hit & out_m
                t_min = torch.where(sel, dist, t_min)
                if minsdf_steps is None:
                    minsdf_steps = torch.empty(p['n_steps']).uniform_(0.0, 1.0)
                dist[m] = min_sdf_search(sdf, origins[m], dirs[m], t_min[m], t_max[m], minsdf_steps, cnt)
        pts = _pts(origins, dist, dirs)
    return {'points': pts, 'hit': hit, 'dists': dist, 'sphere_hit': sph, 'sampler_mask': samp,
            'counters': cnt, 'minsdf_steps': minsdf_steps}
