"""Synthetic workloads for BASELINE.json's configs (no dataset / checkpoint is available).

Everything is generated procedurally from integer seeds with numpy's counter-based Philox
generator, so the same tensors come out in the build container and on the GPU box
(SURVEY.md section 8d).  Model blocks restate the reference's HOCON files:
  physg.conf  code/confs_sg/physg.conf:33-78   (configs 1-2: closed-form SG shading, indirect OFF)
  conf.conf   code/confs_sg/conf.conf:36-94    (configs 3, 5: MC direct + near-field indirect)
  conf_neus   code/confs_sg/conf_neus.conf     (config 4: NeuS geometry, 8x256, d_out 257)
"""
import copy
import math

import numpy as np
import torch

RAY_TRACER = dict(object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=3,
                  sphere_tracing_iters=10, n_steps=100, n_rootfind_steps=32)

PHYSG_MODEL = dict(
    feature_vector_size=0,
    implicit_network=dict(d_in=3, d_out=1, dims=[512] * 8, geometric_init=True, bias=0.6, skip_in=[4],
                          weight_norm=True, multires=6),
    envmap_material_network=dict(multires=10, dims=[512] * 4, white_specular=True, white_light=False,
                                 num_lgt_sgs=128, num_base_materials=1, upper_hemi=False,
                                 fix_specular_albedo=False, specular_albedo=[0.3, 0.3, 0.3]),
    rendering_network=dict(mode='idr', d_in=9, d_out=3, dims=[512] * 4, weight_norm=True, multires_view=4,
                           multires_xyz=10),
    ray_tracer=dict(RAY_TRACER),
)
PHYSG_LOSS = dict(idr_rgb_weight=0.0, sg_rgb_weight=1.0, eikonal_weight=0.1, mask_weight=100.0, alpha=50.0,
                  normalsmooth_weight=1.0, r_patch=1.0, loss_type='L1')

CONF_MODEL = dict(
    render_type='pt_render_indirect_mlp',
    feature_vector_size=512,
    fast_multi_ray=False,
    render_background=True,
    implicit_network=dict(d_in=3, d_out=1, dims=[512] * 8, geometric_init=True, bias=0.6, skip_in=[4],
                          weight_norm=True, multires=6, use_last_as_f=True),
    envmap_material_network=dict(multires=10, dims=[512] * 8, white_specular=True, white_light=False,
                                 num_lgt_sgs=128, num_base_materials=1, upper_hemi=False,
                                 fix_specular_albedo=True, specular_albedo=[0.5, 0.5, 0.5],
                                 init_specular_reflectance=0.1, roughness_mlp=True, specular_mlp=True,
                                 dims_roughness=[512] * 4, dims_specular=[512] * 4, same_mlp=True),
    rendering_network=dict(mode='idr', d_in=9, d_out=3, dims=[512] * 4, weight_norm=True, multires_view=4,
                           multires_xyz=10, normalize_output=False, clip_output=True, clip_method='pow2',
                           weight_init=True),
    ray_tracer=dict(RAY_TRACER),
)
CONF_LOSS = dict(idr_rgb_weight=1.0, sg_rgb_weight=1.0, eikonal_weight=0.1, mask_weight=100.0, alpha=50.0,
                 normalsmooth_weight=1.0, r_patch=1.0, loss_type='L1', env_loss_type='L2',
                 background_rgb_weight=1.0)

NEUS_MODEL = copy.deepcopy(CONF_MODEL)
NEUS_MODEL['feature_vector_size'] = 256
NEUS_MODEL['implicit_network'].update(dims=[256] * 8, bias=0.5, use_last_as_f=False)


def model_conf(name, hidden=None):
    """'physg' | 'conf' | 'neus'; ``hidden`` shrinks every MLP width (small parity-test nets)."""
    m = copy.deepcopy({'physg': PHYSG_MODEL, 'conf': CONF_MODEL, 'neus': NEUS_MODEL}[name])
    if hidden is not None:
        for blk in ('implicit_network', 'envmap_material_network', 'rendering_network'):
            m[blk]['dims'] = [hidden] * len(m[blk]['dims'])
        if m['feature_vector_size'] > 0:
            m['feature_vector_size'] = hidden
    return m


def loss_conf(name):
    return copy.deepcopy(PHYSG_LOSS if name == 'physg' else CONF_LOSS)


def _rng(seed):
    return np.random.Generator(np.random.Philox(int(seed)))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def sdf_layer_dims(cfg, feature_vector_size):
    dims = list(cfg['dims'])
    d0 = 3 + 6 * cfg.get('multires', 0)
    if cfg.get('use_last_as_f', False):
        full = [d0] + dims + [cfg['d_out']]
    else:
        full = [d0] + dims + [cfg['d_out'] + feature_vector_size]
    skip = tuple(cfg.get('skip_in', ()))
    shapes = []
    for l in range(len(full) - 1):
        out = full[l + 1] - full[0] if (l + 1) in skip else full[l + 1]
        shapes.append((out, full[l]))
    return shapes


def fibonacci_sphere(n):
    i = np.arange(n, dtype=np.float64)
    y = 1 - (i / float(n - 1)) * 2
    r = np.sqrt(1 - y * y)
    th = math.pi * (3. - math.sqrt(5.)) * i
    return np.stack([np.cos(th) * r, y, np.sin(th) * r], -1)


SCENES = {'bowl': 'scene_bowl_sdf64.npz', 'bowl_dense': 'scene_bowl_sdf64.npz',
          # TRAINED at the conf's own width by the Step-1 runner (tools/train_scene_sdf.py on a GPU): full-rank weights, no
          # embedding - the file holds the network itself (weight_v in halves, weight_g / bias in fp32)
          'bowl_trained': 'scene_bowl_sdf%d.npz', 'frame_trained': 'scene_frame_sdf%d.npz'}


def _embed_dense(small, shapes, d0, skip, g):
    """The 8 x 64 stand-in at full width WITHOUT zero weights: every small hidden unit is replicated over b / s positions of the
    wide layer (random assignment; incoming weights and bias copied), and its outgoing weights are split over the replicas
    with random positive shares that add up to one - sum_k (a_k w) h = w h: the same function (to fp32 rounding), with every
    weight of every hidden layer non-zero and every activation live.  The zero-padded embedding ('bowl') multiplies 98 % zeros:
    the matrix cores then draw far less power than on a trained network and the part clocks higher."""
    nl = len(shapes)
    eff = []
    for l in range(nl):
        v, gg, b = (small['lin%d.%s' % (l, k)].astype(np.float64) for k in ('weight_v', 'weight_g', 'bias'))
        eff.append((v * (gg / np.linalg.norm(v, axis=1, keepdims=True)), b))
    maps, shares = [], []
    for l in range(nl - 1):
        b_w, s_w = shapes[l][0], eff[l][0].shape[0]
        m = np.empty(b_w, dtype=np.int64)
        m[g.permutation(b_w)] = np.arange(b_w) % s_w
        u = g.uniform(0.5, 1.5, size=b_w)
        tot = np.zeros(s_w)
        np.add.at(tot, m, u)
        maps.append(m)
        shares.append(u / tot[m])
    out = []
    for l, (o, i) in enumerate(shapes):
        ws, b = eff[l]
        rows = maps[l] if l < nl - 1 else np.zeros(1, dtype=np.int64)       # the SDF output itself is not replicated
        if l == 0:
            w = ws[rows]
        else:
            mi, a = maps[l - 1], shares[l - 1]
            if l in skip:
                w = np.concatenate([ws[rows][:, :ws.shape[1] - d0][:, mi] * a, ws[rows][:, ws.shape[1] - d0:]], axis=1)
            else:
                w = ws[rows][:, mi] * a
        out.append((w, b[rows]))
    return out


def embed_scene_sdf(model, sd, scene, g):
    """Overwrite the SDF network of state dict `sd` with the fitted 8 x 64 stand-in geometry `scene`
    (tools/fit_scene_sdf.py: a ball resting in a tilted bowl - a NON-convex body, so that secondary rays re-hit the
    surface and the indirect branch does work), embedded in the conf's own width by padding (SURVEY.md section 8d,
    config 3): the small network's effective weights fill the leading rows / columns of every layer, the encoded input of
    the skip layer keeps its place at the end of the concatenation, the remaining rows get weight_g = 0 with a tiny
    non-zero weight_v (a zero row would make g*v/|v| NaN) and bias 0, the remaining columns are zero.  Same function,
    full-size compute; the padded hidden units all output Softplus(0)."""
    import os
    ic = model['implicit_network']
    F = int(model['feature_vector_size'])
    shapes = sdf_layer_dims(ic, F)
    if scene.endswith('_trained'):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets', SCENES[scene] % ic['dims'][0])
        if not os.path.exists(path):
            raise FileNotFoundError('%s: no trained stand-in for a %d-wide SDF network (tools/train_scene_sdf.py writes it)'
                                    % (path, ic['dims'][0]))
        net = np.load(path)
        for l, (o, i) in enumerate(shapes):
            v = net['lin%d.weight_v' % l].astype(np.float32)
            assert v.shape == (o, i), (scene, l, v.shape, (o, i))
            sd['implicit_network.lin%d.weight_v' % l] = _t(v)
            sd['implicit_network.lin%d.weight_g' % l] = _t(net['lin%d.weight_g' % l].reshape(o, 1))
            sd['implicit_network.lin%d.bias' % l] = _t(net['lin%d.bias' % l])
        return sd
    small = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets', SCENES[scene])))
    d0 = shapes[0][1]
    skip = tuple(ic.get('skip_in', ()))
    nl = len(shapes)
    assert nl == 9 and d0 == 39 and skip == (4,), 'the stand-in was fitted for the 8-layer PE6 skip-4 architecture'
    if scene.endswith('_dense'):
        for l, ((o, i), (w, bias)) in enumerate(zip(shapes, _embed_dense(small, shapes, d0, skip, g))):
            if l == nl - 1 and o > 1:       # NeuS-style feature outputs (conf_neus.conf): random rows, zero bias
                w = np.concatenate([w, g.normal(0.0, math.sqrt(2) / math.sqrt(i), size=(o - 1, i))], axis=0)
                bias = np.concatenate([bias, np.zeros(o - 1)])
            assert w.shape == (o, i), (l, w.shape, (o, i))
            w32 = w.astype(np.float32)
            sd['implicit_network.lin%d.weight_v' % l] = _t(w32)
            sd['implicit_network.lin%d.weight_g' % l] = _t(np.linalg.norm(w32, axis=1, keepdims=True))
            sd['implicit_network.lin%d.bias' % l] = _t(bias)
        return sd
    for l, (o, i) in enumerate(shapes):
        v, gg, b = (small['lin%d.%s' % (l, k)].astype(np.float64) for k in ('weight_v', 'weight_g', 'bias'))
        ws = v * (gg / np.linalg.norm(v, axis=1, keepdims=True))          # effective small weight [o_s, i_s]
        o_s, i_s = ws.shape
        w = np.zeros((o, i))
        if l in skip:
            w[:o_s, :i_s - (d0)] = ws[:, :i_s - d0]
            w[:o_s, i - d0:] = ws[:, i_s - d0:]
        else:
            w[:o_s, :i_s] = ws
        bias = np.zeros((o,))
        bias[:o_s] = b
        if l == nl - 1 and o > 1:       # NeuS-style feature outputs (conf_neus.conf): random rows, zero bias
            w[1:] = g.normal(0.0, math.sqrt(2) / math.sqrt(i), size=(o - 1, i))
            o_s = o
        if o_s < o:
            w[o_s:] = g.normal(0.0, 1e-3, size=(o - o_s, i))
        w32 = w.astype(np.float32)
        wg = np.linalg.norm(w32, axis=1, keepdims=True)
        wg[o_s:] = 0.0
        sd['implicit_network.lin%d.weight_v' % l] = _t(w32)
        sd['implicit_network.lin%d.weight_g' % l] = _t(wg)
        sd['implicit_network.lin%d.bias' % l] = _t(bias)
    return sd


def make_state_dict(model, seed=0, radius=None, bumpy=0.0, scene=None):
    """Procedural weights with the reference's state-dict keys.

    ``scene`` (e.g. 'bowl'): the SDF network is the fitted non-convex stand-in instead (embed_scene_sdf); the
    radiance / material / light parameters are generated exactly as without it.

    ``bumpy`` > 0 puts N(0, bumpy) weights on the sin/cos columns of the first SDF layer: a
    bumpy, non-exact distance field that exercises the tracer's back-off line search,
    sampler and bisection.

    SDF: geometric-init statistics (implicit_differentiable_renderer.py:62-76) -> roughly a
    sphere of radius ``bias``; weight_g = row norm of weight_v.  Radiance / material nets:
    He-normal hidden layers.  lgtSGs as sg_envmap_material.py:134-149."""
    g = _rng(seed)
    F = int(model['feature_vector_size'])
    sd = {}
    ic = model['implicit_network']
    shapes = sdf_layer_dims(ic, F)
    d0 = shapes[0][1]
    bias = ic['bias'] if radius is None else radius
    nl = len(shapes)
    for l, (o, i) in enumerate(shapes):
        if l == nl - 1:
            w = g.normal(math.sqrt(math.pi) / math.sqrt(i), 1e-4, size=(o, i))
            b = np.full((o,), -bias)
            if o > 1:          # NeuS-style feature outputs: small random rows, zero bias
                w[1:] = g.normal(0.0, math.sqrt(2) / math.sqrt(i), size=(o - 1, i))
                b[1:] = 0.0
        else:
            w = g.normal(0.0, math.sqrt(2) / math.sqrt(o), size=(o, i))
            b = np.zeros((o,))
            if l == 0 and d0 > 3:
                w[:, 3:] = g.normal(0.0, bumpy, size=(o, d0 - 3)) if bumpy > 0 else 0.0
            if l in tuple(ic.get('skip_in', ())) and d0 > 3:
                w[:, -(d0 - 3):] = 0.0
        sd['implicit_network.lin%d.weight_v' % l] = _t(w)
        sd['implicit_network.lin%d.weight_g' % l] = _t(np.linalg.norm(w.astype(np.float32), axis=1, keepdims=True))
        sd['implicit_network.lin%d.bias' % l] = _t(b)
    rc = model['rendering_network']
    din = rc['d_in'] + F + 6 * rc.get('multires_view', 0) + 6 * rc.get('multires_xyz', 0)
    dims = [din] + list(rc['dims']) + [rc['d_out']]
    for l in range(len(dims) - 1):
        std = math.sqrt(2.0 / dims[l]) if l < len(dims) - 2 else 1.0 / math.sqrt(dims[l])
        w = g.normal(0.0, std, size=(dims[l + 1], dims[l]))
        sd['rendering_network.lin%d.weight_v' % l] = _t(w)
        sd['rendering_network.lin%d.weight_g' % l] = _t(np.linalg.norm(w.astype(np.float32), axis=1, keepdims=True))
        sd['rendering_network.lin%d.bias' % l] = _t(g.normal(0.0, 0.05, size=(dims[l + 1],)))
    mc = model['envmap_material_network']
    din = 3 + 6 * mc.get('multires', 0) + F
    dout = 3 + (1 if (mc.get('roughness_mlp') and mc.get('same_mlp')) else 0)
    dims = [din] + list(mc['dims']) + [dout]
    for l in range(len(dims) - 1):
        std = math.sqrt(2.0 / dims[l]) if l < len(dims) - 2 else 2.0 / math.sqrt(dims[l])
        sd['envmap_material_network.diffuse_albedo_layers.%d.weight' % (2 * l)] = _t(
            g.normal(0.0, std, size=(dims[l + 1], dims[l])))
        sd['envmap_material_network.diffuse_albedo_layers.%d.bias' % (2 * l)] = _t(
            g.normal(0.0, 0.05, size=(dims[l + 1],)))
    M = mc['num_lgt_sgs']
    lgt = g.normal(size=(M, 7))
    lgt[:, -2:] = lgt[:, -3:-2]
    lgt[:, 3:4] = 20. + np.abs(lgt[:, 3:4] * 100.)
    energy = np.abs(lgt[:, 4:]) * 2.0 * math.pi / lgt[:, 3:4] * (1.0 - np.exp(-2.0 * lgt[:, 3:4]))
    lgt[:, 4:] = np.abs(lgt[:, 4:]) / energy.sum(0, keepdims=True) * 2. * math.pi
    lgt[:, :3] = fibonacci_sphere(M)
    sd['envmap_material_network.lgtSGs'] = _t(lgt)
    if mc.get('fix_specular_albedo'):
        sd['envmap_material_network.specular_reflectance'] = _t(np.array(mc['specular_albedo']).reshape(1, 3))
    else:
        sd['envmap_material_network.specular_reflectance'] = _t(np.abs(g.normal(size=(1, 1 if mc.get('white_specular') else 3))))
    if not mc.get('roughness_mlp'):
        sd['envmap_material_network.roughness'] = _t(g.uniform(1.5, 2.0, size=(1, 1)))
    if scene is not None:
        embed_scene_sdf(model, sd, scene, _rng(seed + 1000))
    return sd


def add_sdf_dent(model, sd, center, width=0.01, depth=0.02, sharp=40.0, value_scale=1.0, row0=None):
    """ADVERSARIAL geometry for the tracer's slope bound (tests only): a small, steep, non-eikonal dent added to an SDF network
    whose wide layers have SPARE hidden units (the zero-padded 'bowl' embedding of the 8 x 64 stand-in: rows 64.. of every layer
    are dead) - in place, the same state-dict keys.

        sdf'(x) = value_scale * sdf(x)  -  depth * max(0, hat(x0) + hat(x1) + hat(x2) - 2),      hat(u) = max(0, 1 - |u - c| / width)

    i.e. a pocket on the octahedron sum_i |x_i - c_i| < width (volume 4/3 width^3: 1.3e-6 for width 0.01 - a random sample of the
    bounding sphere does not find it), `depth` deep at its centre, slope depth / width per axis (|grad| up to sqrt(3) depth /
    width = 3.5 for the defaults) - built from Softplus(beta 100) units: layer 0 holds nine ramps sp(sharp (x_i - c_i - {-w, 0,
    w})) (knee 0.01 / sharp), layer 1 one unit sp(kappa (sum of hats - 2)), layers 2.. carry it (a unit with a +1 offset is in
    Softplus's exactly linear range), the last layer subtracts it.  value_scale < 1 turns the base into an UNDER-estimated
    distance (|grad| = value_scale): sphere tracing then does not converge in its ten iterations and the rays go to the bracket
    search - the path the staged search of eval-mode traces takes.  Returns sd."""
    ic = model['implicit_network']
    shapes = sdf_layer_dims(ic, int(model['feature_vector_size']))
    nl = len(shapes)
    skip = tuple(ic.get('skip_in', ()))
    d0 = shapes[0][1]
    key = 'implicit_network.lin%d.%s'
    # spare rows: weight_g == 0 (embed_scene_sdf's padding)
    spare = [torch.nonzero(sd[key % (l, 'weight_g')].reshape(-1) == 0).flatten().tolist() for l in range(nl - 1)]
    assert len(spare[0]) >= 9 and all(len(sp) >= 1 for sp in spare[1:]), 'add_sdf_dent needs spare hidden units (scene "bowl")'
    c = [float(v) for v in center]
    w = float(width)
    kappa = 1.0             # the layer-1 unit's peak value (exactly linear carry needs value + 1 > 0.2: any kappa >= 0 does)

    def put(l, row, cols_vals, bias):
        v = torch.zeros(shapes[l][1])
        for col, val in cols_vals:
            v[col] = val
        sd[key % (l, 'weight_v')][row] = v
        sd[key % (l, 'weight_g')][row] = v.norm()
        sd[key % (l, 'bias')][row] = bias

    # layer 0: ramps r[i][k] = sp(sharp (x_i - c_i - delta_k)), delta = (-w, 0, +w)
    ramp = []
    rows0 = spare[0][:9]
    for i in range(3):
        for k, dl in enumerate((-w, 0.0, w)):
            put(0, rows0[3 * i + k], [(i, sharp)], -sharp * (c[i] + dl))
            ramp.append(rows0[3 * i + k])
    # layer 1: d = sp(kappa (sum_i hat_i - 2)), hat_i = (r[i][0] - 2 r[i][1] + r[i][2]) / (sharp w)
    r1 = spare[1][0]
    cv = []
    for i in range(3):
        for k, coef in enumerate((1.0, -2.0, 1.0)):
            cv.append((ramp[3 * i + k], kappa * coef / (sharp * w)))
    put(1, r1, cv, -2.0 * kappa)
    # layers 2 .. nl-2: carry (value + 1 per layer; the skip layer's input is divided by sqrt(2))
    prev, carried = r1, 0
    for l in range(2, nl - 1):
        r = spare[l][0]
        put(l, r, [(prev, math.sqrt(2.0) if l in skip else 1.0)], 1.0)
        prev, carried = r, carried + 1
    # last layer: sdf' = value_scale * sdf - (depth / kappa) (carry - carried)
    L = nl - 1
    v, g = sd[key % (L, 'weight_v')], sd[key % (L, 'weight_g')]
    eff = v[0] * (g[0] / v[0].norm())
    eff = eff * value_scale
    eff[prev] = -depth / kappa * (math.sqrt(2.0) if L in skip else 1.0)
    sd[key % (L, 'weight_v')][0] = eff
    sd[key % (L, 'weight_g')][0] = eff.norm()
    sd[key % (L, 'bias')][0] = sd[key % (L, 'bias')][0] * value_scale + depth / kappa * carried
    return sd


def add_sdf_ripple(model, sd, amplitude=0.05, band=5, value_scale=0.3):
    """ADVERSARIAL geometry, the global kind (tests only): sdf'(x) = value_scale * sdf(x) + amplitude * sum_i sin(2^band x_i) on a network
    with spare hidden units (the zero-padded 'bowl') - a high-frequency corrugation everywhere, |grad| up to value_scale + sqrt(3)
    amplitude 2^band (2.8 + 0.3 for the defaults), which a random sample of the bounding sphere DOES see.  One layer-0 unit
    sp(amplitude * sum_i sin(2^band x_i) + 1) on the positional-encoding columns (exactly linear: its argument stays above 0.2), carried
    like add_sdf_dent's unit and added by the last layer.  In place; returns sd."""
    ic = model['implicit_network']
    shapes = sdf_layer_dims(ic, int(model['feature_vector_size']))
    nl = len(shapes)
    skip = tuple(ic.get('skip_in', ()))
    assert 0 <= band < int(ic.get('multires', 0)) and 3.0 * amplitude < 0.75
    key = 'implicit_network.lin%d.%s'
    spare = [torch.nonzero(sd[key % (l, 'weight_g')].reshape(-1) == 0).flatten().tolist() for l in range(nl - 1)]
    assert all(len(sp) >= 1 for sp in spare), 'add_sdf_ripple needs spare hidden units (scene "bowl")'

    def put(l, row, cols_vals, bias):
        v = torch.zeros(shapes[l][1])
        for col, val in cols_vals:
            v[col] = val
        sd[key % (l, 'weight_v')][row] = v
        sd[key % (l, 'weight_g')][row] = v.norm()
        sd[key % (l, 'bias')][row] = bias

    # encoding columns: x (3), then per band k: sin (3), cos (3)  (embedder.py:21-31)
    prev = spare[0][-1]         # (the last spare row: add_sdf_dent takes the first ones - both can be applied to one net)
    put(0, prev, [(3 + 6 * band + i, amplitude) for i in range(3)], 1.0)
    carried = 1
    for l in range(1, nl - 1):
        r = spare[l][-1]
        put(l, r, [(prev, math.sqrt(2.0) if l in skip else 1.0)], 1.0)
        prev, carried = r, carried + 1
    L = nl - 1
    v, g = sd[key % (L, 'weight_v')], sd[key % (L, 'weight_g')]
    eff = v[0] * (g[0] / v[0].norm()) * value_scale
    eff[prev] = math.sqrt(2.0) if L in skip else 1.0
    sd[key % (L, 'weight_v')][0] = eff
    sd[key % (L, 'weight_g')][0] = eff.norm()
    sd[key % (L, 'bias')][0] = sd[key % (L, 'bias')][0] * value_scale - carried
    return sd


def look_at_origin_pose(cam_pos):
    """OpenCV-style cam-to-world (x right, y down, z forward) looking at the origin."""
    c = np.asarray(cam_pos, dtype=np.float64)
    z = -c / np.linalg.norm(c)
    up = np.array([0., 1., 0.]) if abs(z[1]) < 0.99 else np.array([1., 0., 0.])
    x = np.cross(z, up)          # right-handed with y pointing down
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    p = np.eye(4)
    p[:3, 0], p[:3, 1], p[:3, 2], p[:3, 3] = x, y, z, c
    return p


def make_inputs(num_pixels, image_hw=(800, 800), focal=1111.0, cam_pos=(0.0, 0.0, 2.4), num_rays=-1, seed=1,
                mask_all=True, rank=0, world_size=1):
    """One training batch: random 2x2 patches (scene_dataset.py:224-251), optional sub-pixel jitter
    shared across pixels (:212-216), contiguous per-rank slice of the patch list (:268-279).

    Returns (model_input dict, ground-truth rgb [1,S,3])."""
    H, W = image_hw
    g = _rng(seed)
    n_patch = num_pixels // 4
    py = g.integers(0, H - 1, size=n_patch)
    px = g.integers(0, W - 1, size=n_patch)
    per = n_patch // world_size
    lo = rank * per
    hi = n_patch if rank == world_size - 1 else lo + per
    py, px = py[lo:hi], px[lo:hi]
    ys = np.stack([py, py, py + 1, py + 1], 1).reshape(-1)
    xs = np.stack([px, px + 1, px, px + 1], 1).reshape(-1)
    uv = np.stack([xs, ys], -1).astype(np.float64)                      # (u=x, v=y)
    S = uv.shape[0]
    if num_rays > 0:
        jit = g.uniform(-0.5, 0.5, size=(1, num_rays, 2))
        uv = uv[:, None, :] + jit
    K = np.eye(4)
    K[0, 0] = K[1, 1] = focal
    K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    rgb = g.uniform(0.0, 1.0, size=(1, S, 3))
    if mask_all:
        mask = np.ones((1, S), dtype=bool)
    else:
        mask = (g.uniform(size=(1, S)) < 0.7)
    inp = {'uv': _t(uv)[None], 'intrinsics': _t(K)[None], 'pose': _t(look_at_origin_pose(cam_pos))[None],
           'object_mask': torch.from_numpy(mask)}
    return inp, _t(rgb)


WORKLOADS = {
    # BASELINE.json configs (1-based index in SURVEY.md section 8d).  `scene`: None = the geometric-init sphere
    # (configs 1-2, convex), 'bowl' = the fitted non-convex stand-in (a ball in a tilted bowl: ~44 % of the primary rays
    # hit, ~50 % of the secondary rays re-hit, so visibility and the indirect branch do real work)
    'cfg1': dict(model='physg', num_pixels=512, image_hw=(64, 64), focal=137.0, cam_pos=(0., 0., 3.0), num_rays=-1,
                 scene=None),
    'cfg2': dict(model='physg', num_pixels=4096, image_hw=(800, 800), focal=1111.0, cam_pos=(0., 0., 2.4), num_rays=-1,
                 scene=None),
    # config 2 with the camera at 1.6 instead of 2.4: the geometric-init surface (radius 0.32-0.54, mean 0.41) then fills
    # ~42 % of the frame - the hit fraction SURVEY.md section 8(d) planned for config 2 (at 2.4 it is 18 %, so 82 % of the
    # rays go through the min-SDF search); a side measurement of bench.py, not a BASELINE config
    'cfg2_near': dict(model='physg', num_pixels=4096, image_hw=(800, 800), focal=1111.0, cam_pos=(0., 0., 1.6), num_rays=-1,
                      scene=None),
    # configs 3-5: the non-convex stand-in (a ball in a tilted bowl) as the conf's OWN network trained at full width by the
    # Step-1 runner ('bowl_trained', round 5: tools/train_scene_sdf.py).  Before: the 8 x 64 fit replicated across the wide
    # layers without zero weights ('bowl_dense', end of round 4: rank-64 structure - its single-pass error, hence coarse_tau,
    # is twice the trained net's, and config 3 takes 6 % longer on it) and, in rounds 2-4, zero-padded ('bowl': 98 % zero
    # weights cost a power-limited part 18 % less time per step than a dense network does: DESIGN 4d)
    'cfg3': dict(model='conf', num_pixels=4096, image_hw=(800, 800), focal=1111.0, cam_pos=(0., 0., 2.4), num_rays=64,
                 scene='bowl_trained'),
    'cfg4': dict(model='neus', num_pixels=8192, image_hw=(800, 800), focal=1111.0, cam_pos=(0., 0., 2.4), num_rays=64,
                 scene='bowl_trained'),
    # config 5: eval-mode full-frame render (render.py: 800 x 800 pixels in raster order, 256 rays per pixel,
    # memory_capacity_level 18, chunks dealt round-robin over the ranks); num_pixels = pixels per frame
    'cfg5': dict(model='conf', num_pixels=640000, image_hw=(800, 800), focal=1111.0, cam_pos=(0., 0., 2.4), num_rays=256,
                 scene='bowl_trained', eval=True, memory_capacity_level=18),
}


def workload_state_dict(name, seed=0, hidden=None, scene=None):
    """(model conf, state dict) of a WORKLOADS entry.  ``scene`` overrides the entry's embedding of the stand-in geometry
    (the fixtures generated by the reference, tests/golden/make_golden.py, hold the zero-padded 'bowl')."""
    w = WORKLOADS[name]
    mc = model_conf(w['model'], hidden=hidden)
    return mc, make_state_dict(mc, seed=seed, scene=(scene or w.get('scene')) if hidden is None else None)


def frame_inputs(image_hw=(800, 800), focal=1111.0, cam_pos=(0.0, 0.0, 2.4), num_rays=-1, rows=None, seed=2):
    """Full-frame render input (render.py:267-283 / scene_dataset.py:149-216 with sampling_idx None): every pixel of the
    image in raster order - or of the rows `rows` = (first, count) - with the shared sub-pixel jitter when num_rays > 0."""
    H, W = image_hw
    r0, nr = (0, H) if rows is None else rows
    ys, xs = np.mgrid[r0:r0 + nr, 0:W]
    uv = np.stack([xs, ys], -1).reshape(-1, 2).astype(np.float64)
    if num_rays > 0:
        uv = uv[:, None, :] + _rng(seed).uniform(-0.5, 0.5, size=(1, num_rays, 2))
    K = np.eye(4)
    K[0, 0] = K[1, 1] = focal
    K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    return {'uv': _t(uv)[None], 'intrinsics': _t(K)[None], 'pose': _t(look_at_origin_pose(cam_pos))[None],
            'object_mask': torch.ones(1, uv.shape[0], dtype=torch.bool)}
