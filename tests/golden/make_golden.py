#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (build container only).

    python tests/golden/make_golden.py

Imports /root/reference/code through ref_shim (stubs for absent I/O libraries, .cuda() made a
no-op, dict-backed conf) and records inputs + reference outputs as small .npz fixtures.  The
fixtures are data only; the reference never travels to the GPU box.  Weights are NOT stored:
they are regenerated procedurally by nefii_amd.synthetic.make_state_dict(seed) on both sides.
Random draws the reference makes (torch.rand in the samplers, Tensor.uniform_ in
minimal_sdf_points) are captured and stored so that the oracle and the HIP path can replay them.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_shim  # noqa: E402

ref_shim.install()
from model.implicit_differentiable_renderer import IDRNetwork  # noqa: E402  (reference)
from model.loss import IDRLoss  # noqa: E402  (reference)
from model import sg_render as ref_sg  # noqa: E402  (reference)
from model.ray_tracing import RayTracing  # noqa: E402  (reference)
from utils import rend_util as ref_rend  # noqa: E402  (reference)
from nefii_amd import synthetic as syn  # noqa: E402

torch.set_num_threads(8)


class Capture:
    """Record every torch.rand / Tensor.uniform_ draw made while active."""

    def __enter__(self):
        self.rand, self.unif, self.unif_pre, self.unif_post = [], [], [], []
        self._rand, self._unif = torch.rand, torch.Tensor.uniform_

        def rand(*a, **k):
            k.pop('device', None)
            r = self._rand(*a, **k)
            self.rand.append(r.clone())
            return r

        def unif(t, *a, **k):
            r = self._unif(t, *a, **k)
            self.unif.append(r.clone())
            (self.unif_post if self.rand else self.unif_pre).append(r.clone())   # before / after the sampler draws
            return r
        torch.rand, torch.Tensor.uniform_ = rand, unif
        return self

    def __exit__(self, *exc):
        torch.rand, torch.Tensor.uniform_ = self._rand, self._unif


def build_ref(model_cfg, sd, freeze=True):
    with contextlib.redirect_stdout(io.StringIO()):
        m = IDRNetwork(ref_shim.Conf(model_cfg))
    m.load_state_dict(sd, strict=True)
    if freeze:
        m.freeze_geometry()
    return m


def npy(d):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = v
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **npy(arrs))
    print('%-28s %7.1f KB' % (name, os.path.getsize(path) / 1024))


def unit(v):
    return v / v.norm(dim=-1, keepdim=True)


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


# ------------------------------------------------------------------------------------------------
def golden_sg_math():
    lam = torch.logspace(-3, 4, 29).reshape(-1, 1).expand(29, 21).reshape(-1, 1)
    cb = torch.linspace(-1, 1, 21).reshape(1, -1).expand(29, 21).reshape(-1, 1)
    hi = ref_sg.hemisphere_int(lam, cb)
    g = gen(5)
    l1, l2 = unit(torch.randn(300, 3, generator=g)), unit(torch.randn(300, 3, generator=g))
    lam1 = torch.rand(300, 1, generator=g) * 50 + 0.01
    lam2 = lam1 * (1 + torch.rand(300, 1, generator=g) * 1000)
    mu1, mu2 = torch.rand(300, 3, generator=g), torch.rand(300, 3, generator=g)
    fl, fla, fmu = ref_sg.lambda_trick(l1, lam1, mu1, l2, lam2, mu2)
    save('sg_math', lam=lam, cos_beta=cb, hemi=hi, l1=l1, l2=l2, lam1=lam1, lam2=lam2, mu1=mu1, mu2=mu2,
         out_lobe=fl, out_lam=fla, out_mu=fmu)


def golden_sg_render():
    mc = syn.model_conf('physg', hidden=64)
    sd = syn.make_state_dict(mc, seed=3)
    g = gen(7)
    N = 256
    normal = unit(torch.randn(N, 3, generator=g))
    view = unit(normal + 0.8 * torch.randn(N, 3, generator=g))      # mostly front-facing, some grazing/back
    albedo = torch.rand(N, 3, generator=g)
    lgt = sd['envmap_material_network.lgtSGs'].clone().requires_grad_(True)
    rough = torch.tensor([[0.35]], requires_grad=True)
    spec = torch.tensor([[0.04, 0.05, 0.06]], requires_grad=True)
    albedo.requires_grad_(True)
    wts = torch.rand(N, 3, generator=g)
    out = ref_sg.render_with_sg(lgt, spec, rough, albedo, normal, view)
    loss = (out['sg_rgb'] * wts).sum()
    ga, gr, gs, gl = torch.autograd.grad(loss, [albedo, rough, spec, lgt])
    save('sg_render', normal=normal, view=view, albedo=albedo, lgt=lgt, rough=rough, spec=spec, wts=wts,
         sg_rgb=out['sg_rgb'], sg_specular_rgb=out['sg_specular_rgb'], sg_diffuse_rgb=out['sg_diffuse_rgb'],
         g_albedo=ga, g_rough=gr, g_spec=gs, g_lgt=gl)


def golden_nets():
    for name, hidden, seed in [('physg', 64, 0), ('conf', 64, 0), ('neus', 64, 0), ('physg', 512, 0), ('conf', 512, 0)]:
        mc = syn.model_conf(name, hidden=hidden)
        sd = syn.make_state_dict(mc, seed=seed, bumpy=0.02)
        m = build_ref(mc, sd)
        m.eval()
        g = gen(11)
        N = 192
        x = unit(torch.randn(N, 3, generator=g)) * torch.rand(N, 1, generator=g) ** (1 / 3)
        view = unit(torch.randn(N, 3, generator=g))
        y = m.implicit_network(x)
        with torch.enable_grad():
            grad = m.implicit_network.gradient(x.clone(), True)[:, 0, :]
        normals = grad / (grad.norm(dim=-1, keepdim=True) + 1e-6)
        feats = y[:, 1:] if mc['feature_vector_size'] > 0 else None
        rgb = m.rendering_network(x, normals, view, feats)
        mat = m.envmap_material_network(x, feats, normals)
        w1, w2 = torch.rand(N, 3, generator=g), torch.rand(N, 3, generator=g)
        loss = (rgb * w1).sum() + (mat['sg_diffuse_albedo'] * w2).sum()
        if mat['sg_roughness'].shape[0] == N:
            loss = loss + mat['sg_roughness'].sum()
        params = [(k, p) for k, p in m.named_parameters() if p.requires_grad and not k.startswith('implicit')
                  and 'lgtSGs' not in k and k not in ('envmap_material_network.roughness',
                                                      'envmap_material_network.specular_reflectance')]
        gr = torch.autograd.grad(loss, [p for _, p in params])
        extra = {}
        for (k, _), gg in zip(params, gr):
            extra['gnorm.' + k] = gg.norm()
            if hidden == 64:
                extra['grad.' + k] = gg
        fs = y[:, 1:][:, ::16] if y.shape[1] > 1 else None
        save('nets_%s_h%d' % (name, hidden), x=x, view=view, sdf=y[:, :1], feat_sub=fs, grad=grad, rgb=rgb,
             albedo=mat['sg_diffuse_albedo'], roughness=mat['sg_roughness'],
             specular=mat['sg_specular_reflectance'], w1=w1, w2=w2, **extra)


def golden_camera():
    inp, _ = syn.make_inputs(256, image_hw=(64, 64), focal=137.0, cam_pos=(0.3, -0.4, 3.0), num_rays=-1, seed=4)
    K = inp['intrinsics'].clone()
    K[0, 0, 1] = 0.7      # non-zero skew
    K[0, 1, 1] = 140.0
    dirs, cam = ref_rend.get_camera_params(inp['uv'], inp['pose'], K)
    g = gen(13)
    o = torch.randn(1, 3, generator=g) * 0.8
    d = unit(torch.randn(1, 300, 3, generator=g))
    t, hit = ref_rend.get_sphere_intersection(o, d, r=1.0)
    save('camera', uv=inp['uv'], pose=inp['pose'], intrinsics=K, dirs=dirs, cam=cam, o=o, d=d, t=t, hit=hit)


def golden_tracer():
    for tag, name, hidden, bumpy, npx in [('smooth_h64', 'physg', 64, 0.0, 512), ('bumpy_h64', 'physg', 64, 0.03, 512),
                                          ('bumpy_h512', 'physg', 512, 0.004, 128), ('neus_h64', 'neus', 64, 0.02, 256)]:
        mc = syn.model_conf(name, hidden=hidden)
        sd = syn.make_state_dict(mc, seed=0, bumpy=bumpy)
        m = build_ref(mc, sd)
        inp, _ = syn.make_inputs(npx, image_hw=(64, 64), focal=100.0, cam_pos=(0.2, 0.1, 2.0), seed=2, mask_all=False)
        dirs, cam = ref_rend.get_camera_params(inp['uv'], inp['pose'], inp['intrinsics'])
        om = inp['object_mask'].reshape(-1)
        sdf = lambda x: m.implicit_network(x)[:, 0]
        res = {}
        for mode in ('eval', 'train'):
            m.ray_tracer.train(mode == 'train')
            m.implicit_network.eval()
            with torch.no_grad(), Capture() as cap:
                torch.manual_seed(17)
                pts, hit, dist = m.ray_tracer(sdf=sdf, cam_loc=cam, object_mask=om, ray_directions=dirs)
            res[mode + '_points'], res[mode + '_hit'], res[mode + '_dists'] = pts, hit, dist
            if mode == 'train':
                assert len(cap.unif) == 1
                res['minsdf_steps'] = cap.unif[0]
        # secondary-style rays: per-ray origins on a shell inside the sphere, batch of N x 1
        g = gen(19)
        N2 = 192
        o2 = unit(torch.randn(N2, 3, generator=g)) * (0.55 + 0.3 * torch.rand(N2, 1, generator=g))
        d2 = unit(torch.randn(N2, 1, 3, generator=g))
        m.ray_tracer.train(True)
        with torch.no_grad(), Capture() as cap:
            torch.manual_seed(23)
            p2, h2, t2 = m.ray_tracer(sdf=sdf, cam_loc=o2, object_mask=torch.ones(N2, dtype=torch.bool), ray_directions=d2)
        steps2 = cap.unif[0] if cap.unif else torch.zeros(0)
        save('tracer_' + tag, cam=cam, dirs=dirs, object_mask=om, o2=o2, d2=d2.reshape(N2, 3), sec_points=p2,
             sec_hit=h2, sec_dists=t2, minsdf_steps2=steps2, **res)


def run_forward(m, inp, train, seed):
    m.train(train)
    ctx = contextlib.nullcontext() if train else torch.no_grad()
    traces = []      # (points, hit mask, dists) of every RayTracing.forward call: [0] primary rays, [1] secondary rays
    hook = m.ray_tracer.register_forward_hook(lambda mod, args, res: traces.append([t.detach().clone() for t in res]))
    try:
        with ctx, Capture() as cap:
            torch.manual_seed(seed)
            out = m(inp)
    finally:
        hook.remove()
    cap.ray_hit = traces[0][1] if traces else None      # per-RAY hit mask (the output dict only has the per-pixel `all`)
    cap.traces = traces
    return out, cap


def golden_forward_and_step():
    for tag, name, npx, nr, bumpy in [('physg', 'physg', 256, -1, 0.02), ('conf', 'conf', 64, 4, 0.02),
                                      ('neus', 'neus', 64, 2, 0.02)]:
        mc = syn.model_conf(name, hidden=64)
        sd = syn.make_state_dict(mc, seed=0, bumpy=bumpy)
        lc = syn.loss_conf(name)
        m = build_ref(mc, sd)
        inp, gt = syn.make_inputs(npx, image_hw=(64, 64), focal=100.0, cam_pos=(0.2, 0.1, 2.0), num_rays=nr, seed=6,
                                  mask_all=(name != 'physg'))
        for mode in ('train', 'eval'):
            out, cap = run_forward(m, inp, mode == 'train', 31)
            rec = {k: v for k, v in out.items() if v is not None}
            rec['ray_hit'] = cap.ray_hit
            if mode == 'train':
                assert len(cap.unif_pre) <= 1 and len(cap.unif_post) <= 1
                if cap.unif_pre:
                    rec['minsdf_steps'] = cap.unif_pre[0]
                if cap.unif_post:
                    rec['minsdf_steps2'] = cap.unif_post[0]
            if name != 'physg':
                assert len(cap.rand) == 7
                rec['uniforms'] = torch.cat([r.reshape(-1, 1) for r in cap.rand], dim=1)
            if mode == 'train':
                with contextlib.redirect_stdout(io.StringIO()):
                    lossf = IDRLoss(**lc)
                lo = lossf(out, {'rgb': gt})
                m.zero_grad()
                lo['loss'].backward()
                for k, v in lo.items():
                    rec['loss.' + k] = v
                for k, p in m.named_parameters():
                    if p.grad is not None:
                        rec['gnorm.' + k] = p.grad.norm()
                        if p.numel() <= 4096:
                            rec['grad.' + k] = p.grad.clone()
            save('forward_%s_%s' % (tag, mode), uv=inp['uv'], pose=inp['pose'], intrinsics=inp['intrinsics'],
                 in_object_mask=inp['object_mask'], rgb_gt=gt, **rec)
        # secondary-consistency entry point (forward_with_point), eval-free: geometry frozen
        if name == 'conf':
            g = gen(29)
            P = unit(torch.randn(24, 1, 3, generator=g)).expand(24, 2, 3) * 0.6
            D = unit(torch.randn(24, 2, 3, generator=g))
            m.train(True)
            with Capture() as cap:
                torch.manual_seed(37)
                o2 = m({'points': P, 'ray_dirs': D}, with_point=True)
            save('forward_point_conf', points=P, ray_dirs=D, idr_rgb_values=o2['idr_rgb_values'],
                 sg_rgb_values=o2['sg_rgb_values'],
                 uniforms=torch.cat([r.reshape(-1, 1) for r in cap.rand], dim=1),
                 minsdf_steps2=(cap.unif[0] if cap.unif else torch.zeros(0)))


def golden_forward_full_width():
    """IDRNetwork.forward + IDRLoss + backward at the confs' FULL network widths (conf.conf: 8x512 SDF with the 512-wide
    feature vector, 8x512 material, 4x512 radiance; conf_neus.conf: 8x256, d_out 257) on the non-convex stand-in scene of
    configs 3-5 (synthetic.make_state_dict(scene='bowl')), multi-ray pixels through the workload's own camera: pins the
    512-/256-wide path to the reference directly (the hidden-64 fixtures above pin the logic)."""
    for tag, wl, npx, nr in [('conf512', 'cfg3', 32, 4), ('neus256', 'cfg4', 32, 4)]:
        w = syn.WORKLOADS[wl]
        mc, sd = syn.workload_state_dict(wl, seed=0, scene='bowl')
        lc = syn.loss_conf(w['model'])
        m = build_ref(mc, sd)
        inp, gt = syn.make_inputs(npx, w['image_hw'], w['focal'], w['cam_pos'], nr, seed=9)
        out, cap = run_forward(m, inp, True, 41)
        rec = {k: v for k, v in out.items() if v is not None}
        rec['ray_hit'] = cap.ray_hit
        assert len(cap.unif_pre) <= 1 and len(cap.unif_post) <= 1 and len(cap.rand) == 7
        if cap.unif_pre:
            rec['minsdf_steps'] = cap.unif_pre[0]
        if cap.unif_post:
            rec['minsdf_steps2'] = cap.unif_post[0]
        rec['uniforms'] = torch.cat([r.reshape(-1, 1) for r in cap.rand], dim=1)
        with contextlib.redirect_stdout(io.StringIO()):
            lossf = IDRLoss(**lc)
        lo = lossf(out, {'rgb': gt})
        m.zero_grad()
        lo['loss'].backward()
        for k, v in lo.items():
            rec['loss.' + k] = v
        for k, p in m.named_parameters():
            if p.grad is not None:
                rec['gnorm.' + k] = p.grad.norm()
                if p.numel() <= 4096:
                    rec['grad.' + k] = p.grad.clone()
        assert rec['secondary_mask'].float().mean() > 0.2          # the indirect branch is exercised
        save('forward_%s_train' % tag, uv=inp['uv'], pose=inp['pose'], intrinsics=inp['intrinsics'],
             in_object_mask=inp['object_mask'], rgb_gt=gt, **rec)


def golden_trainable_geometry():
    """forward_with_uv with geometry NOT frozen (implicit_differentiable_renderer.py:357-393: SampleNetwork, eikonal
    points, grad_theta) + IDRLoss (eikonal term included) + backward into every parameter, the SDF network's too."""
    for tag, nr in [('physg', -1), ('physg_multi', 2)]:
        mc = syn.model_conf('physg', hidden=64)
        sd = syn.make_state_dict(mc, seed=0, bumpy=0.02)
        lc = syn.loss_conf('physg')
        lc['idr_rgb_weight'] = 1.0
        m = build_ref(mc, sd, freeze=False)
        inp, gt = syn.make_inputs(128 if nr < 0 else 64, image_hw=(64, 64), focal=100.0, cam_pos=(0.2, 0.1, 2.0),
                                  num_rays=nr, seed=6, mask_all=False)
        out, cap = run_forward(m, inp, True, 43)
        assert out['grad_theta'] is not None and len(cap.unif) == 2        # min-SDF steps, eikonal points
        rec = {k: v for k, v in out.items() if v is not None}
        rec['ray_hit'] = cap.ray_hit
        rec['minsdf_steps'], rec['eikonal_points'] = cap.unif[0], cap.unif[1]
        with contextlib.redirect_stdout(io.StringIO()):
            lossf = IDRLoss(**lc)
        lo = lossf(out, {'rgb': gt})
        m.zero_grad()
        lo['loss'].backward()
        for k, v in lo.items():
            rec['loss.' + k] = v
        for k, p in m.named_parameters():
            if p.grad is not None:
                rec['gnorm.' + k] = p.grad.norm()
                if p.numel() <= 4096:
                    rec['grad.' + k] = p.grad.clone()
        assert any(k.startswith('gnorm.implicit_network') and v > 0 for k, v in rec.items())
        save('forward_trainable_%s' % tag, uv=inp['uv'], pose=inp['pose'], intrinsics=inp['intrinsics'],
             in_object_mask=inp['object_mask'], rgb_gt=gt, **rec)


def golden_trainable_geometry_mc():
    """The same branch with the Monte-Carlo render type (conf.conf's pt_render_indirect_mlp, diff_geo=False): sampler draws,
    secondary trace, indirect radiance at secondary hits whose FEATURES stay attached to the SDF network
    (path_tracing_render.py:2109-2166), MIS shading differentiable with respect to the trainable normals."""
    mc = syn.model_conf('conf', hidden=64)
    sd = syn.make_state_dict(mc, seed=0, bumpy=0.02)
    lc = syn.loss_conf('conf')
    m = build_ref(mc, sd, freeze=False)
    inp, gt = syn.make_inputs(48, image_hw=(64, 64), focal=100.0, cam_pos=(0.2, 0.1, 2.0), num_rays=2, seed=6, mask_all=False)
    out, cap = run_forward(m, inp, True, 43)
    assert out['grad_theta'] is not None and len(cap.rand) == 7 and len(cap.traces) == 2
    rec = {k: v for k, v in out.items() if v is not None}
    rec['ray_hit'] = cap.ray_hit
    # uniform_ draws ahead of the sampler: [primary min-SDF steps (if any ray needed them)], eikonal points; behind it:
    # [secondary min-SDF steps]
    rec['eikonal_points'] = cap.unif_pre[-1]
    if len(cap.unif_pre) == 2:
        rec['minsdf_steps'] = cap.unif_pre[0]
    if cap.unif_post:
        rec['minsdf_steps2'] = cap.unif_post[0]
    rec['uniforms'] = torch.cat([r.reshape(-1, 1) for r in cap.rand], dim=1)
    rec['sec_points'], rec['sec_hit'], rec['sec_dists'] = cap.traces[1]
    rec['prim_points'], _, rec['prim_dists'] = cap.traces[0]         # per RAY (`points` is the per-pixel mean)
    with contextlib.redirect_stdout(io.StringIO()):
        lossf = IDRLoss(**lc)
    lo = lossf(out, {'rgb': gt})
    m.zero_grad()
    lo['loss'].backward()
    for k, v in lo.items():
        rec['loss.' + k] = v
    for k, p in m.named_parameters():
        if p.grad is not None:
            rec['gnorm.' + k] = p.grad.norm()
            if p.numel() <= 4096:
                rec['grad.' + k] = p.grad.clone()
    assert any(k.startswith('gnorm.implicit_network') and v > 0 for k, v in rec.items())
    save('forward_trainable_conf_mc', uv=inp['uv'], pose=inp['pose'], intrinsics=inp['intrinsics'],
         in_object_mask=inp['object_mask'], rgb_gt=gt, **rec)


if __name__ == '__main__':
    if sys.argv[1:] == ['trainable']:
        golden_trainable_geometry()
        golden_trainable_geometry_mc()
        sys.exit(0)
    if sys.argv[1:] == ['trainable_mc']:
        golden_trainable_geometry_mc()
        sys.exit(0)
    if sys.argv[1:] == ['full_width']:
        golden_forward_full_width()
        sys.exit(0)
    if sys.argv[1:] == ['forward']:
        golden_forward_and_step()
        golden_forward_full_width()
        sys.exit(0)
    golden_sg_math()
    golden_sg_render()
    golden_camera()
    golden_nets()
    golden_tracer()
    golden_forward_and_step()
    golden_forward_full_width()
    golden_trainable_geometry()
    golden_trainable_geometry_mc()
