"""Trainable-geometry branch of IDRNetwork.forward_with_uv (reference code/model/implicit_differentiable_renderer.py
:357-393, code/model/sample_network.py:10-24) - SURVEY.md section 8a row S1 - as a torch SLOW PATH.

No shipped Step-2 script reaches it (they all pass --freeze_geometry), so it gets no kernels of its own: with geometry
trainable the hit point, its normal and its features are functions of the SDF network's weights, and everything
behind them has to be differentiable with respect to its INPUTS (positions, normals), which the fused MLP / shading
kernels of the frozen path deliberately are not.  What runs where:

  * camera rays and the sphere tracer stay on the HIP kernels (under no_grad, as in the reference :343-350; the
    tracer re-packs the SDF weights whenever they changed);
  * the SDF network, its input gradient (create_graph), SampleNetwork, the radiance and material MLPs and the
    closed-form SG shading are torch ops on the GPU (autograd owns the second-order graph);
  * the Monte-Carlo render type (pt_render_indirect_mlp, round 3): sampler and secondary trace on the HIP kernels under
    no_grad, everything that carries a gradient - the SDF value of the light points, the radiance at secondary hits,
    the light sum and the MIS shading sum - as torch ops (mc_render_torch).

Pinned by a reference-generated fixture (tests/golden/make_golden.py:golden_trainable_geometry): outputs, grad_theta,
all loss terms incl. the eikonal one, and the gradients of every parameter - the SDF network's included."""
import math

import torch
import torch.nn.functional as F

TINY_NUMBER = 1e-6


def embed(x, n_freqs):
    """embedder.py:21-31: x, sin(2^k x), cos(2^k x), k < L (the torch embedder of model/embedder.py)"""
    if n_freqs <= 0:
        return x
    from .embedder import get_embedder
    return get_embedder(n_freqs)[0](x)


def _weights(module, n):
    ws, bs = [], []
    for l in range(n):
        lin = getattr(module, 'lin' + str(l))
        ws.append(torch._weight_norm(lin.weight_v, lin.weight_g, 0) if hasattr(lin, 'weight_g') else lin.weight)
        bs.append(lin.bias)
    return ws, bs


def sdf_torch(net, x):
    """ImplicitNetwork.forward (:85-108) in torch ops: [N,3] -> [N, 1+F]."""
    cfg = net.cfg
    e = embed(x, cfg['multires'])
    ws, bs = _weights(net, len(net.specs))
    n = len(ws)
    h, feat = e, None
    for l in range(n):
        if net.use_last_as_f and l == n - 1:
            feat = h
        if l in cfg['skip_in']:
            h = torch.cat([h, e], dim=1) / math.sqrt(2)
        h = F.linear(h, ws[l], bs[l])
        if l < n - 1:
            h = F.softplus(h, beta=100)
    return torch.cat([h, feat], dim=-1) if net.use_last_as_f else h


def sdf_gradient_torch(net, x, create_graph):
    """ImplicitNetwork.gradient (:110-123): d sdf / d x as [N,1,3]; with create_graph the result stays differentiable
    with respect to the weights (the eikonal term and the normals need that)."""
    with torch.enable_grad():
        x = x.requires_grad_(True) if x.is_leaf else x
        if not x.requires_grad:
            x = x.detach().requires_grad_(True)
        y = sdf_torch(net, x)[:, :1]
        g = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=create_graph, retain_graph=create_graph)[0]
    return g.unsqueeze(1)


def radiance_torch(rn, points, normals, view_dirs, feats):
    """RenderingNetwork.forward (:196-241)."""
    cfg = rn.cfg
    p = embed(points, cfg['multires_xyz'])
    v = embed(view_dirs, cfg['multires_view'])
    if cfg['mode'] == 'idr':
        parts = [p, v, normals]
    elif cfg['mode'] == 'no_view_dir':
        parts = [p, normals]
    else:
        parts = [p, v]
    if rn.feature_vector_size > 0:
        parts.append(feats)
    h = torch.cat(parts, dim=-1)
    ws, bs = _weights(rn, len(rn.specs))
    for l in range(len(ws)):
        h = F.linear(h, ws[l], bs[l])
        if l < len(ws) - 1:
            h = torch.relu(h)
    if cfg['normalize_output']:
        return (torch.tanh(h) + 1.0) / 2.0
    if not cfg['clip_output']:
        return h
    m = cfg['clip_method']
    return {'relu': torch.relu, 'abs': torch.abs, 'pow2': lambda t: t * t,
            'relu_init': lambda t: torch.relu(t) + 0.5}[m](h)


def material_torch(mat, points, feats):
    """EnvmapMaterialNetwork.forward (sg_envmap_material.py:357-425) for the same_mlp / global-parameter layouts."""
    x = embed(points, mat.enc[0])
    if mat.feature_vector_size > 0:
        x = torch.cat([x, feats], dim=-1)
    brdf = torch.sigmoid(mat.diffuse_albedo_layers(x))
    albedo = brdf[..., :3]
    off = 3
    if mat.roughness_mlp:
        rough = brdf[..., off:off + 1]
        off += 1
    else:
        rough = torch.sigmoid(mat.roughness)
    if mat.fix_specular_albedo:
        spec = mat.specular_reflectance
    else:
        spec = brdf[..., off:off + 1] if mat.specular_mlp else torch.sigmoid(mat.specular_reflectance)
        if mat.white_specular:
            spec = spec.expand((-1, 3))
    rough = (1 - 0.089) * rough + 0.089
    if mat.fake_roughness:
        rough = 0 * rough + 0.5
    if mat.fake_specular:
        spec = 0 * spec + 0.5
    return {'sg_lgtSGs': mat.get_lgtSGs(), 'sg_specular_reflectance': mat.specular_remap(spec), 'sg_roughness': rough,
            'sg_diffuse_albedo': albedo, 'sg_blending_weights': None}


# ---- closed-form SG shading (sg_render.py:112-295), one base material, every input differentiable -----------------
def _hemisphere_int(lam, cos_beta):
    lam = lam + TINY_NUMBER
    inv = 1.0 / lam
    t = torch.sqrt(lam) * (1.6988 + 10.8438 * inv) / (1.0 + 6.2201 * inv + 10.2415 * inv * inv)
    inv_a = torch.exp(-t)
    pos = (cos_beta >= 0).to(lam.dtype)
    inv_b = torch.exp(-t * torch.clamp(cos_beta, min=0.0))
    s1 = (1.0 - inv_a * inv_b) / (1.0 - inv_a + inv_b - inv_a * inv_b)
    b = torch.exp(t * torch.clamp(cos_beta, max=0.0))
    s2 = (b - inv_a) / ((1.0 - inv_a) * (b + 1.0))
    s = pos * s1 + (1.0 - pos) * s2
    a_b = 2.0 * math.pi / lam * (torch.exp(-lam) - torch.exp(-2.0 * lam))
    a_u = 2.0 * math.pi / lam * (1.0 - torch.exp(-lam))
    return a_b * (1.0 - s) + a_u * s


def _sg_product(lobe1, lam1, mu1, lobe2, lam2, mu2):
    """product of two SGs with lam1 << lam2 (lambda_trick :141-161)"""
    ratio = lam1 / lam2
    dot = torch.sum(lobe1 * lobe2, dim=-1, keepdim=True)
    tmp = torch.sqrt(ratio * ratio + 1.0 + 2.0 * ratio * dot)
    tmp = torch.min(tmp, ratio + 1.0)
    lam3 = lam2 * tmp
    lobes = (ratio / tmp) * lobe1 + (1.0 / tmp) * lobe2
    return lobes, lam3, mu1 * mu2 * torch.exp(lam2 * (tmp - ratio - 1.0))


def _cosine_integral(normal, lobes, lams, mus):
    """integral over the hemisphere of SG(lobes, lams, mus) x clamped cosine, the cosine as the SG pair of :243-252"""
    mu_cos, lam_cos, alpha_cos = 32.7080, 0.0315, 31.7003
    one = torch.ones_like(lams)
    lobe_p, lam_p, mu_p = _sg_product(normal, lam_cos * one, mu_cos * one, lobes, lams, mus)
    d1 = torch.sum(lobe_p * normal, dim=-1, keepdim=True)
    d2 = torch.sum(lobes * normal, dim=-1, keepdim=True)
    return mu_p * _hemisphere_int(lam_p, d1) - mus * alpha_cos * _hemisphere_int(lams, d2)


def render_with_sg_torch(lgtSGs, specular_reflectance, roughness, diffuse_albedo, normal, viewdirs):
    """render_with_sg for K = 1 (global roughness [1,1] and specular [1,3]); tensors are [N, M, .] inside."""
    assert specular_reflectance.shape[0] == 1 and roughness.shape[0] == 1
    N, M = normal.shape[0], lgtSGs.shape[0]
    n = normal.unsqueeze(1).expand(N, M, 3)
    v = viewdirs.unsqueeze(1).expand(N, M, 3)
    lgt = lgtSGs.unsqueeze(0).expand(N, M, 7)
    l_lobe = lgt[..., :3] / (torch.norm(lgt[..., :3], dim=-1, keepdim=True) + TINY_NUMBER)
    l_lam = torch.abs(lgt[..., 3:4])
    l_mu = torch.abs(lgt[..., -3:])
    inv_r4 = 1.0 / (roughness * roughness * roughness * roughness)                  # [1,1]
    b_lam = (2.0 * inv_r4).reshape(1, 1, 1).expand(N, M, 1)
    b_mu = (inv_r4 / math.pi).reshape(1, 1, 1).expand(N, M, 3)
    v_dot_n = torch.clamp(torch.sum(n * v, dim=-1, keepdim=True), min=0.0)
    w_lobe = 2 * v_dot_n * n - v
    w_lobe = w_lobe / (torch.norm(w_lobe, dim=-1, keepdim=True) + TINY_NUMBER)
    w_lam = b_lam / (4 * v_dot_n + TINY_NUMBER)
    half = w_lobe + v
    half = half / (torch.norm(half, dim=-1, keepdim=True) + TINY_NUMBER)
    v_dot_h = torch.clamp(torch.sum(v * half, dim=-1, keepdim=True), min=0.0)
    s = specular_reflectance.reshape(1, 1, 3).expand(N, M, 3)
    fres = s + (1.0 - s) * torch.pow(2.0, -(5.55473 * v_dot_h + 6.8316) * v_dot_h)
    d1 = torch.clamp(torch.sum(w_lobe * n, dim=-1, keepdim=True), min=0.0)
    d2 = torch.clamp(torch.sum(v * n, dim=-1, keepdim=True), min=0.0)
    k = ((roughness + 1.0) * (roughness + 1.0) / 8.0).reshape(1, 1, 1)
    g = (d1 / (d1 * (1 - k) + k + TINY_NUMBER)) * (d2 / (d2 * (1 - k) + k + TINY_NUMBER))
    w_mu = b_mu * (fres * g / (4 * d1 * d2 + TINY_NUMBER))
    f_lobe, f_lam, f_mu = _sg_product(l_lobe, l_lam, l_mu, w_lobe, w_lam, w_mu)
    spec_rgb = torch.clamp(_cosine_integral(n, f_lobe, f_lam, f_mu).sum(dim=1), min=0.0)
    diff = (diffuse_albedo / math.pi).unsqueeze(1).expand(N, M, 3)
    diff_rgb = torch.clamp(_cosine_integral(n, l_lobe, l_lam, l_mu * diff).sum(dim=1), min=0.0)
    return {'sg_rgb': spec_rgb + diff_rgb, 'sg_specular_rgb': spec_rgb, 'sg_diffuse_rgb': diff_rgb,
            'sg_diffuse_albedo': diffuse_albedo}


# ---- Monte-Carlo direct + near-field indirect shading (path_tracing_render.py:1265-1487, diff_geo=False) -----------------
def _light_along(lgt, wi):
    """sum of the light SGs along directions wi [n, 3] (:1406-1414)"""
    axis = lgt[:, :3] / (torch.norm(lgt[:, :3], dim=-1, keepdim=True) + TINY_NUMBER)
    lam, mu = torch.abs(lgt[:, 3:4]), torch.abs(lgt[:, -3:])
    dots = wi @ axis.t()                                                        # [n, M]
    return torch.exp(lam.reshape(1, -1) * (dots - 1.0)) @ mu


def mc_render_torch(model, mat, normals, view_dirs, points):
    """pt_render_indirect_mlp with every input differentiable: the sampler (nefii_mis_sample) and the secondary trace (the
    HIP tracer) run under no_grad as in the reference (:1283-1354); the SDF value of ALL light points (:2112: its feature
    columns feed the radiance network at the secondary hits, attached to the SDF weights), the normals there
    (gradient(no_grad=True): detached), the radiance network, the light sum and the three-sample MIS sum are torch ops."""
    from .. import ops
    from .path_tracing_render import draw_uniforms
    net = model.implicit_network
    lgt, spec = mat['sg_lgtSGs'], mat['sg_specular_reflectance']
    albedo = mat['sg_diffuse_albedo']
    n = normals.shape[0]
    dev = normals.device
    rough = mat['sg_roughness'].expand(n, 1)
    with torch.no_grad():
        uniforms = getattr(model, 'uniforms_override', None)
        uniforms = draw_uniforms(n, dev) if uniforms is None else uniforms.to(dev)
        wi, own, tab = ops.mis_sample(lgt, rough, normals, view_dirs, uniforms)      # [3,n,3], [3,n], [3,n,3]
        origins = points.detach().unsqueeze(0).expand(3, n, 3).reshape(-1, 3)
        sec_pts, sec_hit, sec_dist = model.ray_tracer(sdf=net, cam_loc=origins,
                                                      object_mask=torch.ones(3 * n, dtype=torch.bool, device=dev),
                                                      ray_directions=wi.reshape(-1, 1, 3))
        hidx = torch.nonzero(sec_hit).flatten()
    light_out = sdf_torch(net, sec_pts)                                         # (:2112) all 3 n light points, with grad
    indirect = torch.zeros(3 * n, 3, device=dev)
    if hidx.numel() > 0:
        lp = sec_pts.index_select(0, hidx)
        g = sdf_gradient_torch(net, lp.clone(), create_graph=False)[:, 0, :].detach()
        nrm = g / (torch.norm(g, dim=-1, keepdim=True) + 1e-6)
        vd = -wi.reshape(-1, 3).index_select(0, hidx)
        vd = vd / (torch.norm(vd, dim=-1, keepdim=True) + 1e-6)
        feats = light_out.index_select(0, hidx)[:, 1:] if model.feature_vector_size > 0 else None
        indirect = indirect.index_put((hidx,), radiance_torch(model.rendering_network, lp, nrm, vd, feats))
    vis = (1.0 - sec_hit.to(torch.float32)).reshape(3, n, 1)
    indirect = indirect.reshape(3, n, 3)
    spec_rgb, diff_rgb = 0.0, 0.0
    r4 = (rough * rough) * (rough * rough)
    for i in range(3):
        w = wi[i]
        light = _light_along(lgt, w)
        half = w + view_dirs
        half = half / (torch.norm(half, dim=-1, keepdim=True) + TINY_NUMBER)
        n_h = torch.clamp(torch.sum(normals * half, dim=-1, keepdim=True), min=0.0)
        root = n_h * n_h + (1.0 - n_h * n_h) / r4
        D = 1.0 / (math.pi * r4 * root * root)
        v_h = torch.clamp(torch.sum(view_dirs * half, dim=-1, keepdim=True), min=0.0)
        fres = spec + (1.0 - spec) * torch.pow(2.0, -(5.55473 * v_h + 6.8316) * v_h)
        d1 = torch.clamp(torch.sum(view_dirs * normals, dim=-1, keepdim=True), min=0.0)
        d2 = torch.clamp(torch.sum(w * normals, dim=-1, keepdim=True), min=0.0)
        k = (rough + 1.0) * (rough + 1.0) / 8.0
        G = (d1 / (d1 * (1 - k) + k + TINY_NUMBER)) * (d2 / (d2 * (1 - k) + k + TINY_NUMBER))
        fs = fres * D * G / (4 * d1 * d2 + TINY_NUMBER)
        pdf = own[i].unsqueeze(-1)
        weight = pdf * pdf / torch.clamp((tab[i] * tab[i]).sum(-1, keepdim=True), min=TINY_NUMBER)   # power heuristic (:390-401)
        l_all = light * vis[i] + (1.0 - vis[i]) * indirect[i]
        spec_rgb = spec_rgb + torch.clamp(weight * l_all * fs * d2 / pdf, min=0.0)
        diff_rgb = diff_rgb + torch.clamp(weight * l_all * (albedo / math.pi) * d2 / pdf, min=0.0)
    return {'sg_rgb': spec_rgb + diff_rgb, 'sg_specular_rgb': spec_rgb, 'sg_diffuse_rgb': diff_rgb,
            'sg_diffuse_albedo': albedo, 'secondary_points': sec_pts.reshape(3, n, 3),
            'secondary_mask': sec_hit.reshape(3, n, 1), 'secondary_dir': wi}


def get_rgb_value(model, points, view_dirs):
    """IDRNetwork.get_rbg_value (:529-599) with geometry trainable: three SDF passes like the reference (features,
    gradient with create_graph), torch radiance / material networks, torch closed-form shading."""
    feats = None
    if model.feature_vector_size > 0:
        feats = sdf_torch(model.implicit_network, points)[:, 1:]
    g = sdf_gradient_torch(model.implicit_network, points, create_graph=True)
    normals = g[:, 0, :]
    normals = normals / (torch.norm(normals, dim=-1, keepdim=True) + 1e-6)
    view_dirs = view_dirs / (torch.norm(view_dirs, dim=-1, keepdim=True) + 1e-6)
    ret = {'normals': normals,
           'idr_rgb': radiance_torch(model.rendering_network, points, normals, view_dirs, feats)}
    mat = material_torch(model.envmap_material_network, points, feats)
    if model.render_type == 'sg':
        ret.update(render_with_sg_torch(mat['sg_lgtSGs'], mat['sg_specular_reflectance'], mat['sg_roughness'],
                                        mat['sg_diffuse_albedo'], normals, view_dirs))
    else:
        ret.update(mc_render_torch(model, mat, normals, view_dirs, points))
    ret.update({'sg_roughness': mat['sg_roughness'], 'sg_specular_reflectance': mat['sg_specular_reflectance']})
    return ret


def forward_with_uv(model, input):
    """forward_with_uv (:312-501) when `model.training and not model.state_freeze_geo`."""
    from ..utils import rend_util
    if model.render_type not in ('sg', 'pt_render_indirect_mlp', 'pt_render_indirect_mlp_memsave'):
        raise NotImplementedError('trainable geometry (torch slow path): render_type %r' % model.render_type)
    uv = input['uv']
    object_mask = input['object_mask'].reshape(-1)
    multi = None
    if uv.dim() == 4:
        B, S, R, _ = uv.shape
        multi = (B, S, R)
        uv = uv.reshape(B, S * R, 2)
        object_mask = object_mask.reshape(B, S, 1).expand(B, S, R).reshape(-1)
    ray_dirs, cam_loc = rend_util.get_camera_params(uv, input['pose'], input['intrinsics'])
    batch_size, num_pixels, _ = ray_dirs.shape
    net = model.implicit_network
    with torch.no_grad():
        _, network_object_mask, dists = model.ray_tracer(sdf=net, cam_loc=cam_loc, object_mask=object_mask,
                                                        ray_directions=ray_dirs)
    model.last_ray_hit = network_object_mask       # per-RAY mask of this forward (diagnostics, as in shade_tail)
    points = (cam_loc.unsqueeze(1) + dists.reshape(batch_size, num_pixels, 1) * ray_dirs).reshape(-1, 3)
    sdf_output = sdf_torch(net, points)[:, 0:1]
    ray_dirs = ray_dirs.reshape(-1, 3)
    surface_mask = network_object_mask & object_mask
    sidx = torch.nonzero(surface_mask).flatten()
    surface_points = points.index_select(0, sidx)
    N = surface_points.shape[0]
    surface_dists = dists.index_select(0, sidx).unsqueeze(-1)
    surface_ray_dirs = ray_dirs.index_select(0, sidx)
    surface_cam_loc = cam_loc.unsqueeze(1).expand(batch_size, num_pixels, 3).reshape(-1, 3).index_select(0, sidx)
    surface_output = sdf_output.index_select(0, sidx)
    # points for the eikonal term: uniform in the bounding box + the traced points (:368-374)
    n_eik = batch_size * num_pixels // 2
    bb = model.object_bounding_sphere
    eik = getattr(model, 'eikonal_points_override', None)
    if eik is None:
        eik = torch.empty(n_eik, 3).uniform_(-bb, bb)
    eik = torch.cat([eik.to(points.device), points.detach()], dim=0)
    points_all = torch.cat([surface_points, eik], dim=0)
    surface_sdf_values = sdf_torch(net, surface_points)[:N, 0:1].detach()
    g = sdf_gradient_torch(net, points_all, create_graph=True)
    surface_points_grad = g[:N, 0, :].clone().detach()
    grad_theta = g[N:, 0, :]
    diff_points = model.sample_network(surface_output, surface_sdf_values, surface_points_grad, surface_dists,
                                       surface_cam_loc, surface_ray_dirs)
    n_all = points.shape[0]
    dev = points.device
    out = {'idr_rgb_values': torch.ones(n_all, 3, device=dev), 'sg_rgb_values': torch.ones(n_all, 3, device=dev),
           'normal_values': torch.ones(n_all, 3, device=dev), 'sg_diffuse_rgb_values': torch.ones(n_all, 3, device=dev),
           'sg_diffuse_albedo_values': torch.ones(n_all, 3, device=dev),
           'sg_specular_rgb_values': torch.zeros(n_all, 3, device=dev),
           'sg_roughness_values': torch.zeros(n_all, 1, device=dev),
           'sg_specular_reflection_values': torch.zeros(n_all, 3, device=dev)}
    if N > 0:
        ret = get_rgb_value(model, diff_points, -surface_ray_dirs)
        for key, src in (('idr_rgb_values', 'idr_rgb'), ('sg_rgb_values', 'sg_rgb'), ('normal_values', 'normals'),
                         ('sg_diffuse_rgb_values', 'sg_diffuse_rgb'), ('sg_diffuse_albedo_values', 'sg_diffuse_albedo'),
                         ('sg_specular_rgb_values', 'sg_specular_rgb'), ('sg_roughness_values', 'sg_roughness'),
                         ('sg_specular_reflection_values', 'sg_specular_reflectance')):
            val = ret[src].expand(N, out[key].shape[1])
            out[key] = out[key].index_put((sidx,), val)
    if model.render_background:
        bidx = torch.nonzero(~surface_mask).flatten()
        if bidx.numel() > 0:
            bg = model.get_background_rgb(ray_dirs.index_select(0, bidx))
            out['sg_rgb_values'] = out['sg_rgb_values'].index_put((bidx,), bg)
    output = {'points': points, 'sdf_output': sdf_output, 'network_object_mask': network_object_mask,
              'object_mask': object_mask, 'grad_theta': grad_theta,
              'secondary_points': ret.get('secondary_points') if N > 0 else None,
              'secondary_mask': ret.get('secondary_mask') if N > 0 else None,
              'secondary_dir': ret.get('secondary_dir') if N > 0 else None}
    output.update(out)
    if multi is not None:
        B, S, R = multi
        for key in ['idr_rgb_values', 'sg_rgb_values', 'network_object_mask', 'object_mask', 'sg_diffuse_rgb_values',
                    'sg_diffuse_albedo_values', 'sg_specular_rgb_values', 'sdf_output', 'points', 'sg_roughness_values',
                    'sg_specular_reflection_values']:
            output[key] = model.mean_pixel(output[key], B * S, R)
        output['normal_values'] = model.mean_pixel(output['normal_values'], B * S, R, vector=True)
    return output
