"""CPU study for VERDICT r5 next #5: bisection with the slope bound L (ray_tracing.py:259-280).

After a midpoint m has been evaluated (value v), the root lies at least |v| / L away from it; a following midpoint m' on the
same side within that distance has an implied sign (chained: |f(m')| >= |v| - L |m' - m|) and needs no evaluation.  Both bracket
ends count (their values are known from the bracket search).  Bit-identical to the plain bisection whenever L bounds the slope.

The oracle's own tracer on config 3's trained stand-in: rays that reach the bisection, evaluations with and without the rule,
for L = s x the largest |grad sdf| of the network (s = 1.0: the tightest honest bound; 1.5: what ops.calibrate_lipschitz ships),
plus the distribution of the slope along the ray at the root relative to L - the rule saves a step only where
slope / L > 1/2 (linear model: the next midpoint is half an interval away, the root at most one)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from nefii_amd import synthetic as syn
from oracle import nets, tracer


def bisect_with_bound(sdf, o, d, lo, hi, f_lo, f_hi, n_steps, L):
    """returns (root, evaluations made, evaluations the plain recurrence makes).  Vectorised over rays; anchors a (positive side)
    and b (negative side) hold the last EVALUATED point and value on each side."""
    n = lo.shape[0]
    a_t, a_v, b_t, b_v = lo.clone(), f_lo.clone(), hi.clone(), f_hi.clone()
    lo, hi = lo.clone(), hi.clone()
    mid = (lo + hi) / 2
    work = (f_lo > 0) & (f_hi < 0) & (hi > lo)
    made = torch.zeros(n, dtype=torch.long)
    plain = torch.zeros(n, dtype=torch.long)
    for _ in range(n_steps):
        if not work.any():
            break
        plain += work.long()
        pos_implied = work & (a_v - L * (mid - a_t) > 0)
        neg_implied = work & (b_v + L * (b_t - mid) < 0)
        need = work & ~pos_implied & ~neg_implied
        f_mid = torch.zeros(n)
        if need.any():
            f_mid[need] = sdf(o[need] + mid[need].unsqueeze(-1) * d[need])
            made += need.long()
        up = (need & (f_mid > 0)) | pos_implied
        dn = (need & ~(f_mid > 0)) | neg_implied
        ev_up, ev_dn = need & (f_mid > 0), need & ~(f_mid > 0)
        a_t = torch.where(ev_up, mid, a_t)
        a_v = torch.where(ev_up, f_mid, a_v)
        b_t = torch.where(ev_dn, mid, b_t)
        b_v = torch.where(ev_dn, f_mid, b_v)
        lo = torch.where(up, mid, lo)
        hi = torch.where(dn, mid, hi)
        mid = torch.where(work, (lo + hi) / 2., mid)
        work = work & ((hi - lo) > 1e-6)
    return mid, made, plain


def main():
    torch.set_num_threads(8)
    w = syn.WORKLOADS['cfg3']
    mc, sd = syn.workload_state_dict('cfg3', seed=0)
    cfg = mc['implicit_network']
    sdf = lambda x: nets.sdf_forward(sd, cfg, x)[:, 0]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(60000, 3, generator=g)
    x = x / x.norm(dim=1, keepdim=True) * torch.rand(60000, 1, generator=g) ** (1 / 3.0)
    gmax = nets.sdf_gradient(sd, cfg, x).reshape(-1, 3).norm(dim=1).max().item()
    print('largest |grad sdf| over 60000 points of the bounding sphere: %.3f' % gmax)
    inp, _ = syn.make_inputs(256, w['image_hw'], w['focal'], w['cam_pos'], 16, seed=1)
    B, S, R, _ = inp['uv'].shape
    from oracle import renderer as orr
    Ro = orr.Renderer({k: v.clone() for k, v in sd.items()}, mc, training=False)
    flat = dict(inp)
    flat['uv'] = inp['uv'].reshape(B, S * R, 2)
    dirs, cam = Ro.camera_rays(flat) if hasattr(Ro, 'camera_rays') else (None, None)
    if dirs is None:
        from oracle import shading  # noqa: F401
        import math
        # pinhole rays of the workload's camera (synthetic.make_inputs' intrinsics / pose)
        uv = flat['uv'][0]
        K, pose = flat['intrinsics'][0], flat['pose'][0]
        xc = (uv[:, 0] - K[0, 2]) / K[0, 0]
        yc = (uv[:, 1] - K[1, 2]) / K[1, 1]
        dcam = torch.stack([xc, yc, torch.ones_like(xc)], -1)
        dirs = torch.nn.functional.normalize(dcam @ pose[:3, :3].t(), dim=1)
        cam = pose[:3, 3].expand_as(dirs)
    o, d = cam.contiguous(), dirs.contiguous()
    p = dict(tracer.DEFAULT_TRACER)
    p.update(mc['ray_tracer'])
    # primary rays, then secondary-like rays from the primary hit points
    res = tracer.trace(sdf, o, d, torch.ones(o.shape[0], dtype=torch.bool), mc['ray_tracer'], False)
    hp = res['points'][res['hit']][:3000]
    nrm = torch.nn.functional.normalize(nets.sdf_gradient(sd, cfg, hp).reshape(-1, 3), dim=1)
    w2 = torch.nn.functional.normalize(torch.randn(hp.shape[0], 3, generator=g), dim=1)
    w2 = torch.where((w2 * nrm).sum(1, keepdim=True) < 0, -w2, w2)
    for what, (oo, dd) in (('primary', (o, d)), ('secondary', (hp + 0.0 * w2, w2))):
        cnt = tracer.Counters()
        t_io, sph = tracer.sphere_intersection(oo, dd, p['object_bounding_sphere'])
        live, t_s, t_e, t_min, t_max = tracer.sphere_trace(sdf, oo, dd, sph, t_io, p, cnt)
        m = live
        if not m.any():
            print(what, 'no ray reaches the bracket search')
            continue
        n = p['n_steps']
        lin = torch.linspace(0, 1, steps=n)
        ts = t_s[m].unsqueeze(-1) + lin.view(1, -1) * (t_e[m] - t_s[m]).unsqueeze(-1)
        vals = sdf((oo[m].unsqueeze(1) + ts.unsqueeze(-1) * dd[m].unsqueeze(1)).reshape(-1, 3)).reshape(-1, n)
        ind = tracer.first_crossing(vals)
        rows = torch.arange(vals.shape[0])
        hit = vals[rows, ind] < 0
        r = rows[hit]
        hi, f_hi = ts[r, ind[hit]], vals[r, ind[hit]]
        lo, f_lo = ts[r, ind[hit] - 1], vals[r, ind[hit] - 1]
        ob, db = oo[m][hit], dd[m][hit]
        ref = tracer.bisect(sdf, ob, db, lo.clone(), hi.clone(), f_lo, f_hi, p, tracer.Counters())
        slope = ((f_lo - f_hi) / (hi - lo)).abs()
        print('%s: %d of %d rays in the bracket search, %d reach the bisection; slope along the ray over the bracket: median %.2f, '
              '90th pct %.2f, max %.2f' % (what, int(m.sum()), oo.shape[0], r.numel(), slope.median(), slope.quantile(0.9), slope.max()))
        for s in (1.0, 1.5):
            L = s * gmax
            root, made, plain = bisect_with_bound(sdf, ob, db, lo, hi, f_lo, f_hi, p['n_rootfind_steps'], L)
            same = '%s (max |d root| %.1e: the CPU GEMM rounds differently for different batch shapes)' % (torch.equal(root, ref), (root - ref).abs().max().item())
            print('   L = %.1f x %.3f = %.3f: evaluations %d -> %d (x %.3f), roots bit-identical to the plain bisection: %s; rays with '
                  'slope / L > 0.5: %.3f' % (s, gmax, L, plain.sum(), made.sum(), made.sum().item() / max(plain.sum().item(), 1), same,
                                               (slope / L > 0.5).float().mean()))


if __name__ == '__main__':
    main()
