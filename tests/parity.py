"""Shared comparison helpers of the GPU parity tests (renderer level).

north_star tolerance: relative L2 <= 1e-3 on rendered RGB (sg_rgb_values) and albedo (sg_diffuse_albedo_values) on
identical rays.  Monte-Carlo shaded colours are compared UNTRIMMED on every ray whose discrete sampling events are the
same on both sides; rays with a proven discrete difference - a sampled direction that differs (the SG-mixture sampler
picks its lobe by a CDF comparison: a uniform within rounding noise of a boundary picks the neighbouring lobe) or a
secondary ray whose hit flag differs (a grazing re-hit decided by an `sdf <= 5e-5` comparison) - are counted, bounded
and reported, never silently dropped."""
import os

import torch

# Soft mode (the parity PROTOCOL of an arithmetic that is not expected to meet every bound, tools/tier_round.sh): a tolerance that
# does not hold is printed ("EXCEEDS") instead of raised, so that one run lists every figure.  It is switched on by the explicit
# pytest option `--parity-soft` only (tests/conftest.py sets this flag and prints a banner); no environment variable reaches it -
# a leaked NEFII_PARITY_SOFT fails the session at collection instead of making the suite pass vacuously.
SOFT = False


def _check(ok, info):
    if ok:
        return
    if SOFT:
        print('[parity EXCEEDS] %s' % (info,))
    else:
        raise AssertionError(info)


FLOAT_KEYS = ['points', 'idr_rgb_values', 'sg_rgb_values', 'normal_values', 'sdf_output', 'sg_diffuse_rgb_values',
              'sg_diffuse_albedo_values', 'sg_specular_rgb_values', 'sg_roughness_values',
              'sg_specular_reflection_values']
MC_KEYS = ('sg_rgb_values', 'sg_diffuse_rgb_values', 'sg_specular_rgb_values')


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _spread(x, hit):
    """[3, N_hit, C] rows of the hit rays -> [3, N_ray, C] (zeros on rays that miss)"""
    x = x.detach().cpu()
    full = torch.zeros(3, hit.shape[0], x.shape[-1], dtype=x.dtype)
    full[:, hit] = x
    return full


def mc_flagged_rays(out, ref, hit, rhit, dir_tol=1e-3):
    """bool [N_ray]: primary rays both sides hit whose Monte-Carlo sampling differs DISCRETELY between `out` and `ref` -
    one of the 3 sampled directions differs by more than dir_tol, or one of the 3 secondary rays hits on one side only.
    Returns (flagged, n_direction, n_visibility)."""
    hit, rhit = hit.cpu().bool(), rhit.cpu().bool()
    both = hit & rhit
    d = (_spread(out['secondary_dir'], hit) - _spread(ref['secondary_dir'], rhit)).abs().amax(-1) > dir_tol      # [3, N]
    m = _spread(out['secondary_mask'].float(), hit)[..., 0] != _spread(ref['secondary_mask'].float(), rhit)[..., 0]
    dflag, mflag = d.any(0) & both, m.any(0) & both & ~d.any(0)
    return dflag | mflag, int(dflag.sum()), int(mflag.sum())


NORTH_STAR_KEYS = ('sg_rgb_values', 'sg_diffuse_albedo_values')


def compare_outputs(out, ref, tol_rgb=1e-3, max_flips=1, what='', rays_per_pixel=1, ray_hit=None, ref_ray_hit=None,
                    max_explained_frac=0.0, sdf_outliers=0, tol_aux=None, tol_points=1e-4, miss_sdf_max=5e-3):
    """`out` (HIP path) against `ref` (oracle output or reference-generated fixture), per pixel.

    ray_hit / ref_ray_hit: the per-ray hit masks of both sides (model.last_ray_hit, oracle '_ray_hit' / fixture
    'ray_hit'); with them, pixels containing a ray with a discrete Monte-Carlo difference (mc_flagged_rays) are excluded
    from the three MC-shaded colour keys - at most max_explained_frac of the pixels - and every other pixel is compared
    untrimmed.  tol_aux (default tol_rgb): the bound of every key OTHER than the two the north star names (rendered RGB, albedo)
    - the tiered sphere tracing moves hit points by up to ~sdf_threshold / cos, and the random-weight material network of the
    synthetic workloads (PE10: 2^9 x position) turns that into 2-3e-3 on the roughness channel.
    miss_sdf_max: bound on |sdf_output| differences of rays that MISS on both sides; None = report only.  In eval mode a missing
    ray's point is wherever its two sphere-tracing fronts passed each other - nothing reads it - and with the tiered sphere
    tracing the fronts advance by single-pass values: that crossing place moves."""
    tol_aux = tol_rgb if tol_aux is None else tol_aux
    net, rnet = out['network_object_mask'].cpu(), ref['network_object_mask']
    flips = (net != rnet).sum().item()
    _check(flips <= max_flips, (what, 'hit-mask flips', flips))
    assert torch.equal(out['object_mask'].cpu(), ref['object_mask'])
    agree = net == rnet
    flagged_px = torch.zeros_like(agree)
    n_dir = n_vis = 0
    if ray_hit is not None and out.get('secondary_dir') is not None and ref.get('secondary_dir') is not None:
        flagged, n_dir, n_vis = mc_flagged_rays(out, ref, ray_hit, ref_ray_hit)
        flagged_px = flagged.reshape(-1, rays_per_pixel).any(1)
        frac = flagged_px.float().mean().item()
        _check(frac <= max_explained_frac, (what, 'pixels with a discrete MC difference', frac, n_dir, n_vis))
    print('[parity %s] pixels %d, hit-mask flips %d, rays with a differing sampled direction %d / secondary hit flag %d '
          '-> %d pixels compared apart' % (what, net.numel(), flips, n_dir, n_vis, int(flagged_px.sum())))
    figures = []
    for k in FLOAT_KEYS:
        keep = agree & ~flagged_px if k in MC_KEYS else agree
        a, b = out[k].detach().cpu()[keep], ref[k][keep]
        figures.append('%s %.2e' % (k.replace('_values', ''), rel_l2(a[rnet[keep]], b[rnet[keep]]) if k in ('points', 'sdf_output')
                                   else rel_l2(a, b)))
        if k in ('points', 'sdf_output'):
            # rays that miss take the argmin of 100 samples (flat minimum: the winner flips on rounding noise);
            # compare them on hit rays only, misses through sdf_output (the value reached) with a loose bound
            h = rnet[keep]
            if k == 'points':
                figures.append('|d point| max %.2e' % ((a[h] - b[h]).abs().max().item() if h.any() else 0.0))
                _check(rel_l2(a[h], b[h]) < tol_points, (what, k, rel_l2(a[h], b[h])))
            else:      # |sdf| <= 5e-5 on the surface: absolute comparison (sdf_outliers: hit rays allowed beyond it -
                # large samples contain the odd ray whose bisection bracket differs by one sample)
                _check(int(((a[h] - b[h]).abs() >= 2e-4).sum()) <= sdf_outliers, (what, k, (a[h] - b[h]).abs().max().item()))
                _check((a[h] - b[h]).abs().median().item() < 2e-6, (what, k))
            if k == 'sdf_output' and (~h).any():
                figures.append('|d sdf| of missing rays max %.2e' % (a[~h] - b[~h]).abs().max().item())
                figures.append('within 2e-5: %.3f' % ((a[~h] - b[~h]).abs() < 2e-5).float().mean().item())
                if miss_sdf_max is not None:
                    _check((a[~h] - b[~h]).abs().max().item() < miss_sdf_max, (what, k))
                    _check(((a[~h] - b[~h]).abs() < 2e-5).float().mean().item() > 0.9, (what, k))
            continue
        if k == 'sg_specular_rgb_values':
            # A COMPONENT of the rendered colour: GGX's D = 1 / (pi a^2 ((n.h)^2 + (1 - (n.h)^2) / a^2)^2), a = roughness^2,
            # amplifies the fp32 rounding of n.h by up to 1 / a^2 (path_tracing_render.py:1428-1434), so on low-roughness
            # pixels two fp32 evaluations of the reference's own formula differ by ~1e-3 of the lobe's peak.  The north-star
            # bound is on the rendered RGB (sg_rgb_values, checked above at tol_rgb): the specular part must stay within
            # tol_rgb at THAT scale, and within 10 tol_rgb of its own.
            total = ref['sg_rgb_values'][keep]
            err = (a.float() - b.float()).norm().item()
            _check(err / (total.norm().item() + 1e-12) < tol_rgb, (what, k, 'vs rgb', err / total.norm().item()))
            _check(rel_l2(a, b) < 10 * tol_aux, (what, k, rel_l2(a, b)))
            continue
        _check(rel_l2(a, b) < (tol_rgb if k in NORTH_STAR_KEYS else tol_aux), (what, k, rel_l2(a, b)))
    print('[parity %s] rel-L2: %s' % (what, ', '.join(figures)))
    # the excluded pixels are not unchecked: everything that does not pass through the sampler still has to agree there
    # (above: normals, albedo, roughness, idr_rgb over all `agree` pixels), and their colours must stay finite
    for k in MC_KEYS:
        assert torch.isfinite(out[k]).all(), (what, k)
    keep = agree & ~flagged_px
    return {'flips': flips, 'dir': n_dir, 'vis': n_vis, 'flagged_pixels': int(flagged_px.sum()),
            # what the bound was held to: rendered RGB over the pixels whose Monte-Carlo samples are the same on both sides
            'rgb_rel_l2_same_samples': rel_l2(out['sg_rgb_values'].detach().cpu()[keep], ref['sg_rgb_values'][keep])}
