"""Oracle: the three MLPs of the path (functional, driven by a state_dict).

TEST INFRASTRUCTURE (see oracle/__init__.py).  fp32 PyTorch-CPU restatement of
  * positional encoding                 reference code/model/embedder.py:5-50
  * ImplicitNetwork.forward / gradient  code/model/implicit_differentiable_renderer.py:85-123
  * RenderingNetwork.forward            code/model/implicit_differentiable_renderer.py:196-241
  * EnvmapMaterialNetwork.forward       code/model/sg_envmap_material.py:357-425
All functions take the reference's state_dict keys (weight-norm
``lin{l}.weight_g/.weight_v/.bias``; ``diffuse_albedo_layers.{2i}.weight/.bias``).
"""
import math

import torch
import torch.nn.functional as F


def posenc(x, n_freqs):
    """[N,3] -> [N, 3+6L]: x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x).

    embedder.py:21-31 (log-sampled bands 2^0..2^(L-1), include_input=True)."""
    if n_freqs <= 0:
        return x
    parts = [x]
    for k in range(n_freqs):
        f = float(2 ** k)
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, dim=-1)


def linear_params(sd, prefix):
    """Effective (W [out,in], b [out]) of one layer; undoes torch weight_norm (dim=0)."""
    if prefix + '.weight_g' in sd:
        g = sd[prefix + '.weight_g']
        v = sd[prefix + '.weight_v']
        w = v * (g / v.norm(dim=1, keepdim=True))
    else:
        w = sd[prefix + '.weight']
    return w, sd[prefix + '.bias']


def count_layers(sd, prefix, fmt='lin{}'):
    n = 0
    while (prefix + '.' + fmt.format(n) + '.bias') in sd:
        n += 1
    return n


def sdf_forward(sd, cfg, x, prefix='implicit_network'):
    """ImplicitNetwork.forward: [N,3] -> [N, 1+F].

    cfg: the ``implicit_network`` conf block (+ 'feature_vector_size').
    Skip layers concatenate the encoded input and divide by sqrt(2)
    (:97-98); Softplus(beta=100) after all but the last layer (:102-103);
    use_last_as_f appends the last hidden activation as the feature (:92-93,105-106)."""
    skip_in = tuple(cfg.get('skip_in', ()))
    last_as_f = bool(cfg.get('use_last_as_f', False))
    enc = posenc(x, int(cfg.get('multires', 0)))
    n_lin = count_layers(sd, prefix)
    h = enc
    feat = None
    for l in range(n_lin):
        if last_as_f and l == n_lin - 1:
            feat = h
        if l in skip_in:
            h = torch.cat([h, enc], dim=1) / math.sqrt(2)
        w, b = linear_params(sd, '%s.lin%d' % (prefix, l))
        h = F.linear(h, w, b)
        if l < n_lin - 1:
            h = F.softplus(h, beta=100)
    if last_as_f:
        h = torch.cat([h, feat], dim=-1)
    return h


def sdf_gradient(sd, cfg, x, prefix='implicit_network'):
    """d sdf / d x, [N,3]  (ImplicitNetwork.gradient :110-123, no_grad=True flavour)."""
    with torch.enable_grad():
        xr = x.detach().clone().requires_grad_(True)
        y = sdf_forward(sd, cfg, xr, prefix)[:, :1]
        g = torch.autograd.grad(y, xr, torch.ones_like(y))[0]
    return g.detach()


def radiance_forward(sd, cfg, points, normals, view_dirs, feats, prefix='rendering_network'):
    """RenderingNetwork.forward (mode 'idr'): cat[PE(x), PE(v), n, feat] -> ReLU MLP -> head.

    Head: tanh -> (x+1)/2 when normalize_output (default, :228-230); x**2 for
    clip_method 'pow2' (:240-241); relu/abs/relu_init likewise."""
    mode = cfg.get('mode', 'idr')
    v = posenc(view_dirs, int(cfg.get('multires_view', 0)))
    p = posenc(points, int(cfg.get('multires_xyz', 0)))
    if mode == 'idr':
        parts = [p, v, normals]
    elif mode == 'no_view_dir':
        parts = [p, normals]
    elif mode == 'no_normal':
        parts = [p, v]
    else:
        raise ValueError(mode)
    if feats is not None:
        parts.append(feats)
    h = torch.cat(parts, dim=-1)
    n_lin = count_layers(sd, prefix)
    for l in range(n_lin):
        w, b = linear_params(sd, '%s.lin%d' % (prefix, l))
        h = F.linear(h, w, b)
        if l < n_lin - 1:
            h = torch.relu(h)
    if cfg.get('normalize_output', True):
        return (torch.tanh(h) + 1.) / 2.
    if not cfg.get('clip_output', False):
        return h
    m = cfg.get('clip_method', 'relu')
    if m == 'relu':
        return torch.relu(h)
    if m == 'abs':
        return torch.abs(h)
    if m == 'relu_init':
        return torch.relu(h) + 0.5
    if m == 'pow2':
        return h ** 2
    raise ValueError(m)


TINY_ROUGHNESS = 0.089   # sg_envmap_material.py:403


def material_forward(sd, cfg, points, feats, prefix='envmap_material_network',
                     fake_roughness=False, fake_specular=False):
    """EnvmapMaterialNetwork.forward for the two shipped flavours.

    conf.conf   : same_mlp, roughness_mlp, specular_mlp, fix_specular_albedo ->
                  MLP out 4 = albedo(3)+roughness(1); specular = fixed param [1,3].
    physg.conf  : MLP out 3 = albedo; global ``roughness`` [1,1] and
                  ``specular_reflectance`` [1,1] (white_specular) through sigmoid.
    Returns dict with the reference's keys (:418-425)."""
    h = posenc(points, int(cfg.get('multires', 0)))
    if feats is not None:
        h = torch.cat([h, feats], dim=-1)
    lp = prefix + '.diffuse_albedo_layers'
    idx = sorted({int(k[len(lp) + 1:].split('.')[0]) for k in sd if k.startswith(lp + '.')})
    for j, i in enumerate(idx):
        h = F.linear(h, sd['%s.%d.weight' % (lp, i)], sd['%s.%d.bias' % (lp, i)])
        if j < len(idx) - 1:
            h = F.elu(h)
    albedo = torch.sigmoid(h[..., :3])
    roughness_mlp = bool(cfg.get('roughness_mlp', False))
    specular_mlp = bool(cfg.get('specular_mlp', False))
    same_mlp = bool(cfg.get('same_mlp', False))
    fix_spec = bool(cfg.get('fix_specular_albedo', False))
    white_spec = bool(cfg.get('white_specular', False))
    if (roughness_mlp or specular_mlp) and not same_mlp:
        raise NotImplementedError('separate roughness/specular MLPs are not used by any shipped conf')
    off = 3
    if roughness_mlp:
        rough = torch.sigmoid(h[..., off:off + 1])
        off += 1
    else:
        rough = torch.sigmoid(sd[prefix + '.roughness'])
    if fix_spec:
        spec = sd[prefix + '.specular_reflectance']
    else:
        if specular_mlp:
            spec = torch.sigmoid(h[..., off:off + 1])
            off += 1
        else:
            spec = torch.sigmoid(sd[prefix + '.specular_reflectance'])
        if white_spec:
            spec = spec.expand(-1, 3)
    rough = (1 - TINY_ROUGHNESS) * rough + TINY_ROUGHNESS
    if fake_roughness:
        rough = 0 * rough + 0.5
    if fake_specular:
        spec = 0 * spec + 0.5
    spec = 0.16 * spec ** 2           # specular_remap :442-443
    lgt = sd[prefix + '.lgtSGs']
    if cfg.get('white_light', False):
        lgt = torch.cat((lgt, lgt[..., -1:], lgt[..., -1:]), dim=-1)
    if cfg.get('upper_hemi', False):
        lgt = torch.cat((lgt[..., :1], torch.abs(lgt[..., 1:2]), lgt[..., 2:]), dim=-1)
    return {'sg_lgtSGs': lgt, 'sg_specular_reflectance': spec, 'sg_roughness': rough,
            'sg_diffuse_albedo': albedo, 'sg_blending_weights': None}
