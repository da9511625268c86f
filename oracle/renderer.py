"""Oracle: camera rays, IDRNetwork.forward orchestration, loss and one optimiser step.

TEST INFRASTRUCTURE (see oracle/__init__.py).  fp32 PyTorch-CPU restatement of
  * get_camera_params / lift            code/utils/rend_util.py:90-142
  * IDRNetwork.forward_with_uv          code/model/implicit_differentiable_renderer.py:312-501
  * IDRNetwork.forward_with_point       code/model/implicit_differentiable_renderer.py:503-527
  * IDRNetwork.get_rbg_value            code/model/implicit_differentiable_renderer.py:529-599
  * get_visibility_and_indirect_light   code/model/path_tracing_render.py:2109-2166
  * pt_render_indirect_mlp              code/model/path_tracing_render.py:1255-1487 (diff_geo=False)
  * IDRLoss.forward                     code/model/loss.py:162-320 (terms with non-zero weight in shipped confs)
Geometry is frozen (Step-2 ``--freeze_geometry``): the non-frozen branch
(:357-393, SampleNetwork) is outside the path's scope.
"""
import torch
import torch.nn.functional as F

from . import nets, shading, tracer


def camera_rays(uv, pose, intrinsics):
    """uv [B,S,2] pixel coords, pose [B,4,4] cam-to-world, K [B,4,4] -> unit dirs [B,S,3], cam [B,3]."""
    cam = pose[:, :3, 3]
    fx = intrinsics[:, 0, 0].unsqueeze(-1)
    fy = intrinsics[:, 1, 1].unsqueeze(-1)
    cx = intrinsics[:, 0, 2].unsqueeze(-1)
    cy = intrinsics[:, 1, 2].unsqueeze(-1)
    sk = intrinsics[:, 0, 1].unsqueeze(-1)
    x = uv[:, :, 0]
    y = uv[:, :, 1]
    z = torch.ones_like(x)
    xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    yl = (y - cy) / fy * z
    hom = torch.stack((xl, yl, z, torch.ones_like(z)), dim=-1)          # [B,S,4]
    world = torch.bmm(pose, hom.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
    dirs = F.normalize(world - cam[:, None, :], dim=2)
    return dirs, cam


def _unit(v):
    return v / (torch.norm(v, dim=-1, keepdim=True) + 1e-6)


class Renderer:
    """Functional counterpart of IDRNetwork driven by (state_dict, model conf dict)."""

    def __init__(self, sd, model_cfg, training=True):
        self.sd = sd
        self.cfg = model_cfg
        self.F = int(model_cfg['feature_vector_size'])
        self.sdf_cfg = dict(model_cfg['implicit_network'])
        self.rad_cfg = dict(model_cfg['rendering_network'])
        self.mat_cfg = dict(model_cfg['envmap_material_network'])
        self.trc_cfg = dict(model_cfg['ray_tracer'])
        self.render_type = model_cfg.get('render_type', 'sg')
        self.render_background = bool(model_cfg.get('render_background', False))
        self.training = training
        self.fake_roughness = False
        self.counters = tracer.Counters()
        self.dead_work = True      # execute the reference's redundant SDF passes (cpu baseline fidelity)

    # -- pieces -------------------------------------------------------------------------------
    def sdf(self, x):
        self.counters.add('sdf_calls', x.shape[0])
        return nets.sdf_forward(self.sd, self.sdf_cfg, x)

    def sdf0(self, x):
        return self.sdf(x)[:, 0]

    def trace(self, origins, dirs, object_mask, minsdf_steps=None):
        return tracer.trace(self.sdf0, origins, dirs, object_mask, self.trc_cfg,
                            self.training, minsdf_steps, self.counters)

    def surface_terms(self, pts):
        """(features or None, unit normals) at surface points (get_rbg_value :531-541)."""
        feats = None
        if self.F > 0:
            feats = self.sdf(pts)[:, 1:].detach()
        g = nets.sdf_gradient(self.sd, self.sdf_cfg, pts)
        self.counters.add('sdf_grad', pts.shape[0])
        return feats, _unit(g)

    def radiance(self, pts, normals, view, feats):
        return nets.radiance_forward(self.sd, self.rad_cfg, pts, normals, view, feats)

    def shade(self, pts, view, uniforms=None, minsdf_steps2=None):
        feats, normals = self.surface_terms(pts)
        view = _unit(view)
        idr_rgb = self.radiance(pts, normals, view, feats)
        mat = nets.material_forward(self.sd, self.mat_cfg, pts, feats, fake_roughness=self.fake_roughness)
        ret = {'normals': normals, 'idr_rgb': idr_rgb}
        if self.render_type == 'sg':
            ret.update(shading.sg_closed_form(mat['sg_lgtSGs'], mat['sg_specular_reflectance'], mat['sg_roughness'],
                                              mat['sg_diffuse_albedo'], normals, view))
        elif self.render_type in ('pt_render_indirect_mlp', 'pt_render_indirect_mlp_memsave'):
            ret.update(self.shade_indirect(mat, pts, normals, view, uniforms, minsdf_steps2))
        else:
            raise NotImplementedError(self.render_type)
        ret['sg_roughness'] = mat['sg_roughness']
        ret['sg_specular_reflectance'] = mat['sg_specular_reflectance']
        return ret

    def shade_indirect(self, mat, pts, normals, view, uniforms, minsdf_steps2):
        lgt = mat['sg_lgtSGs']
        rough = mat['sg_roughness']
        ws, own, table, uniforms = shading.draw_mis_directions(lgt.detach(), rough.detach(), normals, view, uniforms)
        n = pts.shape[0]
        o3 = pts.detach().repeat(3, 1)
        d3 = torch.cat(ws, dim=0)
        tr = self.trace(o3, d3, torch.ones(3 * n, dtype=torch.bool), minsdf_steps2)
        vis, ind = [], []
        for i in range(3):
            sl = slice(i * n, (i + 1) * n)
            hit = tr['hit'][sl]
            lp = tr['points'][sl]
            if self.dead_work:
                self.sdf(lp)           # reference evaluates the SDF on all light points (:2112)
            rgb = torch.zeros(n, 3)
            if hit.any():
                lps = lp[hit]
                feats, nrm = self.surface_terms(lps)
                rgb = rgb.index_put((torch.nonzero(hit).flatten(),),
                                    self.radiance(lps, nrm, _unit(-ws[i][hit]), feats))
            vis.append(1 - hit.float().unsqueeze(-1))
            ind.append(rgb)
        out = shading.mc_shade(lgt, mat['sg_specular_reflectance'], rough, mat['sg_diffuse_albedo'],
                               normals, view, ws, own, table, vis, ind)
        out['secondary_points'] = tr['points'].reshape(3, n, 3)
        out['secondary_mask'] = tr['hit'].reshape(3, n, 1)
        out['secondary_dir'] = torch.stack(ws, dim=0)
        out['_uniforms'] = uniforms
        out['_minsdf_steps2'] = tr['minsdf_steps']
        return out

    # -- forward_with_uv ------------------------------------------------------------------------
    def forward(self, inp, minsdf_steps=None, uniforms=None, minsdf_steps2=None):
        """uniforms: the sampler's 7 draws per HIT ray [N_hit, 7] as the reference makes them, or per ray [N_ray, 7]
        (rows of rays that miss are ignored) for callers that fix the draws before knowing which rays hit."""
        uv = inp['uv']
        object_mask = inp['object_mask'].reshape(-1)
        multi = uv.dim() == 4
        if multi:
            B, S, R, _ = uv.shape
            uv = uv.reshape(B, S * R, 2)
            object_mask = object_mask.reshape(B, S, 1).expand(B, S, R).reshape(-1)
        dirs, cam = camera_rays(uv, inp['pose'], inp['intrinsics'])
        B, P, _ = dirs.shape
        o = cam.unsqueeze(1).expand(B, P, 3).reshape(-1, 3)
        d = dirs.reshape(-1, 3)
        tr = self.trace(o, d, object_mask, minsdf_steps)
        hit, dists = tr['hit'], tr['dists']
        pts = (cam.unsqueeze(1) + dists.reshape(B, P, 1) * dirs).reshape(-1, 3)
        sdf_out = self.sdf(pts)[:, 0:1].detach()
        ones = torch.ones_like(pts)
        out = {k: ones.clone() for k in ('idr_rgb_values', 'sg_rgb_values', 'normal_values',
                                         'sg_diffuse_rgb_values', 'sg_diffuse_albedo_values')}
        out['sg_specular_rgb_values'] = torch.zeros_like(pts)
        out['sg_roughness_values'] = torch.zeros_like(pts[:, :1])
        out['sg_specular_reflection_values'] = torch.zeros_like(pts)
        ret = {}
        if hit.any():
            idx = torch.nonzero(hit).flatten()
            if uniforms is not None and uniforms.shape[0] == hit.shape[0]:
                uniforms = uniforms[hit]
            ret = self.shade(pts[hit], -d[hit], uniforms, minsdf_steps2)
            put = lambda dst, src: dst.index_put((idx,), src.expand(idx.shape[0], dst.shape[1]))
            out['idr_rgb_values'] = put(out['idr_rgb_values'], ret['idr_rgb'])
            out['sg_rgb_values'] = put(out['sg_rgb_values'], ret['sg_rgb'])
            out['normal_values'] = put(out['normal_values'], ret['normals'])
            out['sg_diffuse_rgb_values'] = put(out['sg_diffuse_rgb_values'], ret['sg_diffuse_rgb'])
            out['sg_diffuse_albedo_values'] = put(out['sg_diffuse_albedo_values'], ret['sg_diffuse_albedo'])
            out['sg_specular_rgb_values'] = put(out['sg_specular_rgb_values'], ret['sg_specular_rgb'])
            out['sg_roughness_values'] = put(out['sg_roughness_values'], ret['sg_roughness'])
            out['sg_specular_reflection_values'] = put(out['sg_specular_reflection_values'],
                                                       ret['sg_specular_reflectance'])
        miss = ~hit
        if self.render_background and miss.any():
            bg = shading.env_radiance(self.sd['envmap_material_network.lgtSGs'], d[miss])
            out['sg_rgb_values'] = out['sg_rgb_values'].index_put((torch.nonzero(miss).flatten(),), bg)
        out.update({'points': pts, 'sdf_output': sdf_out, 'network_object_mask': hit,
                    'object_mask': object_mask, 'grad_theta': None,
                    'secondary_points': ret.get('secondary_points'), 'secondary_mask': ret.get('secondary_mask'),
                    'secondary_dir': ret.get('secondary_dir')})
        out['_minsdf_steps'] = tr['minsdf_steps']
        out['_ray_hit'] = hit            # per-RAY mask (network_object_mask becomes the per-pixel `all` below)
        out['_uniforms'] = ret.get('_uniforms')
        out['_minsdf_steps2'] = ret.get('_minsdf_steps2')
        if multi:
            n = B * S
            for k in ('idr_rgb_values', 'sg_rgb_values', 'sg_diffuse_rgb_values', 'sg_diffuse_albedo_values',
                      'sg_specular_rgb_values', 'sdf_output', 'points', 'sg_roughness_values',
                      'sg_specular_reflection_values'):
                out[k] = out[k].reshape(n, R, -1).mean(1)
            out['network_object_mask'] = out['network_object_mask'].reshape(n, R).all(1)
            out['object_mask'] = out['object_mask'].reshape(n, R).all(1)
            out['normal_values'] = out['normal_values'].reshape(n, R, 3)[:, 0, :]
        return out

    def forward_with_point(self, points, ray_dirs):
        """points, ray_dirs [N,R,3] -> mean-over-R idr / sg rgb (:503-527)."""
        N, R, _ = points.shape
        ret = self.shade(points.reshape(-1, 3), -ray_dirs.reshape(-1, 3))
        return {'idr_rgb_values': ret['idr_rgb'].reshape(N, R, 3).mean(1),
                'sg_rgb_values': ret['sg_rgb'].reshape(N, R, 3).mean(1)}


def idr_loss(out, rgb_gt, lc):
    """IDRLoss.forward restricted to terms the shipped confs weight (loss.py:278-320)."""
    net, obj = out['network_object_mask'], out['object_mask']
    gt = rgb_gt.reshape(-1, 3)
    zero = torch.tensor(0.0)
    m = net & obj
    img = F.l1_loss if lc.get('loss_type', 'L1') == 'L1' else F.mse_loss
    idr_l, sg_l = zero, zero
    if m.any():
        idr_l = img(out['idr_rgb_values'][m], gt[m])
        sg_l = img(out['sg_rgb_values'][m], gt[m])
    mm = ~m
    mask_l = zero
    if mm.any():
        alpha = lc['alpha']
        mask_l = (1 / alpha) * F.binary_cross_entropy_with_logits(
            (-alpha * out['sdf_output'][mm]).squeeze(-1), obj[mm].float(), reduction='sum') / float(obj.shape[0])
    ns_l = zero
    rp = int(lc.get('r_patch', -1))
    if rp >= 1 and lc.get('normalsmooth_weight', 0.) != 0.:
        pm = m.reshape(-1, 4 * rp * rp).all(dim=-1)
        if pm.any():
            ns_l = torch.mean(torch.var(out['normal_values'].view(-1, 4 * rp * rp, 3), dim=1)[pm])
    bg_l = zero
    bgm = (~net) & (~obj)
    if lc.get('background_rgb_weight', 0.) > 0 and bgm.any():
        env = F.l1_loss if lc.get('env_loss_type', 'L1') == 'L1' else F.mse_loss
        bg_l = env(out['sg_rgb_values'][bgm], gt[bgm])
    loss = (lc['idr_rgb_weight'] * idr_l + lc['sg_rgb_weight'] * sg_l + lc['mask_weight'] * mask_l +
            lc.get('normalsmooth_weight', 0.) * ns_l + lc.get('background_rgb_weight', 0.) * bg_l)
    return {'loss': loss, 'idr_rgb_loss': idr_l, 'sg_rgb_loss': sg_l, 'mask_loss': mask_l,
            'normalsmooth_loss': ns_l, 'background_rgb_loss': bg_l}
