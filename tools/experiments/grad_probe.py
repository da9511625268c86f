"""Per-parameter gradient parity of config 3's shrunk workload (tests/test_gpu_configs.py::test_config_shrunk_in_pixels_vs_oracle)
against the CPU oracle: relative L2 of EVERY parameter gradient, for both embeddings of the stand-in geometry (zero-padded
'bowl', replicated 'bowl_dense') and three arithmetics of the radiance / material MLPs' training path -
    f16x3 + half state (the default: split-precision forward, one-pass fp16 backward, stash and dz in halves)
    f16x3 + fp32 state (NEFII_MLP_H16=0)
    f32             (NEFII_MLP_PRECISION=f32: the f32-input MFMA kernels, bit-exact fp32 fma chains)
- so that the bound of the test (3e-3 on 'bowl', 6e-3 on 'bowl_dense') can be read against which parameters approach it and
why: VERDICT r4 weak #1.  The table goes to profiles/r05/grad_probe_cfg3.txt.   python tools/experiments/grad_probe.py [pixels]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch

from nefii_amd import synthetic as syn
from oracle import renderer as orr
import test_gpu_configs as T

px = int(sys.argv[1]) if len(sys.argv) > 1 else T.SHRUNK['cfg3']
w = syn.WORKLOADS['cfg3']
lc = syn.loss_conf(w['model'])
inp, gt = syn.make_inputs(px, w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1)
flat, gt_flat, R = T.per_ray_layout(inp, gt)
n_ray = flat['uv'].shape[1]
g = torch.Generator().manual_seed(5)
steps1, steps2 = torch.rand(100, generator=g), torch.rand(100, generator=g)
uniforms = torch.rand(n_ray, 7, generator=g)
VARIANTS = [('f16x3, half state', {}), ('f16x3, fp32 state', {'NEFII_MLP_H16': '0'}), ('f32 MLP kernels', {'NEFII_MLP_PRECISION': 'f32'})]
table, names, notes = {}, None, []
for scene in ('bowl', 'bowl_dense'):
    mc, sd = syn.workload_state_dict('cfg3', seed=0, scene=scene)
    sdo = {k: v.clone() for k, v in sd.items()}
    for k in sdo:
        if not k.startswith('implicit') and not (k.endswith('specular_reflectance') and mc['envmap_material_network'].get('fix_specular_albedo')):
            sdo[k].requires_grad_(True)
    Ro = orr.Renderer(sdo, mc, training=True)
    Ro.dead_work = False
    ref = Ro.forward(flat, steps1, uniforms, steps2)
    orr.idr_loss(ref, gt_flat, lc)['loss'].backward()
    # the same oracle in fp64: how far is the fp32 oracle itself from it?  (the gradients of the radiance net's first layers
    # are sums with heavy cancellation: summation order alone moves them by 1-2.4e-3 in fp32)
    torch.set_default_dtype(torch.float64)
    try:
        sd64 = {k: v.clone().double() for k, v in sd.items()}
        for k in sd64:
            if sdo[k].requires_grad:
                sd64[k].requires_grad_(True)
        R64 = orr.Renderer(sd64, mc, training=True)
        R64.dead_work = False
        f64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in flat.items()}
        ref64 = R64.forward(f64, steps1.double(), uniforms.double(), steps2.double())
        orr.idr_loss(ref64, gt_flat.double(), lc)['loss'].backward()
    finally:
        torch.set_default_dtype(torch.float32)
    same_rays = torch.equal(ref['_ray_hit'], ref64['_ray_hit']) and torch.equal(ref['secondary_mask'], ref64['secondary_mask'])
    table[(scene, 'fp32 oracle')] = {n: (T.rel_l2(sdo[n].grad, sd64[n].grad.float()), float(sd64[n].grad.norm()))
                                     for n in sdo if sdo[n].grad is not None and sd64[n].grad is not None and sd64[n].grad.norm() > 0}
    notes.append('%s: fp32 and fp64 oracle trace the same rays: %s' % (scene, same_rays))
    for tag, env in VARIANTS:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            from nefii_amd.model.loss import IDRLoss
            m = T.build_model(mc, sd, True)
            m.secondary_miss_search = True
            m.ray_tracer.minsdf_steps_override = [steps1, steps2]
            out = T.gpu_forward_with_per_ray_draws(m, T.to_dev(flat), uniforms)
            IDRLoss(**lc)(out, {'rgb': gt_flat.to(T.DEV)})['loss'].backward()
            col, col64 = {}, {}
            for name, p in m.named_parameters():
                gref = sdo[name].grad
                if gref is not None and gref.norm() > 0 and p.grad is not None:
                    col[name] = (T.rel_l2(p.grad, gref), float(gref.norm()))
                    col64[name] = (T.rel_l2(p.grad, sd64[name].grad.float()), float(gref.norm()))
            table[(scene, tag)] = col
            table[(scene, tag + ' /64')] = col64
            names = names or list(col)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
cols = [(s, t) for s in ('bowl', 'bowl_dense') for t in ['fp32 oracle'] + [v[0] for v in VARIANTS] + [VARIANTS[0][0] + ' /64']]
print('config 3, %d pixels x %d rays: relative L2 of every parameter gradient against the CPU oracle (fp32); column "fp32 oracle": the '
      'fp32 oracle against the SAME oracle in fp64; columns "/64": the HIP path against the fp64 oracle' % (px, w['num_rays']))
print('\n'.join(notes))
print('%-58s %10s | ' % ('parameter', '|grad|') + ' | '.join('%-10s %-17s' % c for c in cols))
for n in names:
    print('%-58s %10.2e | ' % (n, table[cols[0]][n][1]) + ' | '.join('%28.2e' % table[c].get(n, (float('nan'),))[0] for c in cols))
print('%-58s %10s | ' % ('WORST', '') + ' | '.join('%28.2e' % max(v[0] for v in table[c].values()) for c in cols))
for c in cols:
    over = sorted(((v[0], n) for n, v in table[c].items() if v[0] > 2e-3), reverse=True)[:8]
    print('%-10s %-17s above 2e-3: %s' % (c[0], c[1], ', '.join('%s %.2e' % (n, v) for v, n in over) or 'none'))
