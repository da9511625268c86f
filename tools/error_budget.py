"""The 1e-3 north-star bound on rendered RGB / albedo, split over its consumers (VERDICT r5 next #3a): the HIP path against the CPU
oracle on identical rays, weights and draws, at the per-ray sizes bench.py times (config 3 / 4: 128 pixels x 64 rays of the
training batch; config 5: 32 pixels x 256 rays, eval mode), with the arithmetic switched on one consumer at a time:

    f32        exact-fp32 kernels everywhere (f32-input MFMA tracer and surface pass, f32 MLPs): the floor - summation order only
    +split     the tracer's evaluators as shipped: split precision (3 fp16 MFMAs per product) + the bit-identical coarse / staged passes,
               the split-precision surface pass (value, features, normals)
    +fp16mlp   the radiance / material MLPs on fp16 tiles (split-precision forward; this is the library default, untiered)
    +tier      tiered sphere tracing (RayTracing.trace_tier: what bench.py runs)
    [+fp8corr  the split evaluator's correction products on block-scaled fp8 (NEFII_SPLIT_FP8=1), when the library has it]

Prints one table row per (workload, arithmetic): hit-mask flips, RGB rel-L2 over all hit pixels and over the pixels whose Monte-Carlo
samples are the same on both sides, albedo rel-L2, rays with another sampled lobe.  One oracle forward per workload (cached).

    python tools/error_budget.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    dev = torch.device('cuda', 0)
    torch.set_num_threads(16)
    rows = []
    fp8 = os.environ.get('NEFII_BUDGET_FP8', '1') == '1'
    variants = [('f32', dict(mlp='f32', tracer='f32', tier=False)), ('+split', dict(mlp='f32', tracer='f16x3w', tier=False)),
                ('+fp16mlp', dict(mlp='f16x3', tracer='f16x3w', tier=False)), ('+tier', dict(mlp='f16x3', tracer='f16x3w', tier=True))]
    if fp8:
        variants.append(('+fp8corr', dict(mlp='f16x3', tracer='f16x3w', tier=True, fp8=True)))
    for wl, pp in (('cfg2', 512), ('cfg3', 128), ('cfg4', 128), ('cfg5', 64)):
        ref = bench.oracle_reference(wl, pp)
        for name, v in variants:
            os.environ['NEFII_MLP_PRECISION'] = v['mlp']
            if v.get('fp8'):
                os.environ['NEFII_SPLIT_FP8'] = '1'
            try:
                def tweak(m, v=v):
                    m.ray_tracer.precision = v['tracer']
                p = bench.parity_of(wl, ref, dev, v['tier'], tweak=tweak)
            finally:
                os.environ.pop('NEFII_MLP_PRECISION', None)
                os.environ.pop('NEFII_SPLIT_FP8', None)
            row = {'workload': wl, 'arithmetic': name, 'pixels': p['pixels'], 'hit_pixels': p['hit_pixels'],
                   'flips': p['hit_mask_mismatches'], 'ray_flips': p.get('ray_hit_mismatches'), 'rgb_rel_l2': p['rgb_rel_l2'], 'albedo_rel_l2': p['albedo_rel_l2'],
                   'rgb_rel_l2_same_samples': p.get('rgb_rel_l2_same_samples'),
                   'rays_with_another_sampled_direction': p.get('rays_with_another_sampled_direction'),
                   'rays_with_another_secondary_hit_flag': p.get('rays_with_another_secondary_hit_flag')}
            rows.append(row)
            print('%-5s %-9s hit pixels %3d/%3d flips %d (rays: %s) | RGB %.2e (same samples %s) albedo %.2e | rays with another lobe %s, another '
                  'secondary hit flag %s' % (wl, name, row['hit_pixels'], row['pixels'], row['flips'], row['ray_flips'], row['rgb_rel_l2'],
                                             '%.2e' % row['rgb_rel_l2_same_samples'] if row['rgb_rel_l2_same_samples'] is not None else '-',
                                             row['albedo_rel_l2'], row['rays_with_another_sampled_direction'],
                                             row['rays_with_another_secondary_hit_flag']), flush=True)
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
