"""Oracle: spherical-Gaussian shading (closed form, and 3-sample MIS Monte-Carlo).

TEST INFRASTRUCTURE (see oracle/__init__.py).  fp32 PyTorch-CPU restatement of
  * hemisphere_int / lambda_trick / render_with_sg   code/model/sg_render.py:112-295
  * samplers + pdfs                                  code/model/path_tracing_render.py:12-271
  * MIS weights, sg_fn                               code/model/path_tracing_render.py:390-413
  * pt_render_diff_shadow_indirect_mlp (diff_geo=False) shading sum
                                                     code/model/path_tracing_render.py:1406-1476
  * background SG evaluation                         code/model/implicit_differentiable_renderer.py:646-663
Tensors are laid out [N, M, 3] (point, lobe, channel); the number of base
materials K is 1 in every shipped conf and is dropped.
"""
import math

import torch

TINY = 1e-6
PI = math.pi


def unpack_light(lgt):
    """lgtSGs [M,7] -> unit lobe axes [M,3], |lambda| [M,1], |mu| [M,3] (sg_render.py:194-196)."""
    axis = lgt[..., :3] / (torch.norm(lgt[..., :3], dim=-1, keepdim=True) + TINY)
    return axis, torch.abs(lgt[..., 3:4]), torch.abs(lgt[..., -3:])


def hemi_integral(lam, cos_b):
    """Clamped-hemisphere integral of an SG, rational fit (sg_render.py:112-138)."""
    lam = lam + TINY
    inv = 1. / lam
    t = torch.sqrt(lam) * (1.6988 + 10.8438 * inv) / (1. + 6.2201 * inv + 10.2415 * inv * inv)
    ia = torch.exp(-t)
    pos = (cos_b >= 0).float()
    ib = torch.exp(-t * torch.clamp(cos_b, min=0.))
    s1 = (1. - ia * ib) / (1. - ia + ib - ia * ib)
    b = torch.exp(t * torch.clamp(cos_b, max=0.))
    s2 = (b - ia) / ((1. - ia) * (b + 1.))
    s = pos * s1 + (1. - pos) * s2
    a_b = 2. * PI / lam * (torch.exp(-lam) - torch.exp(-2. * lam))
    a_u = 2. * PI / lam * (1. - torch.exp(-lam))
    return a_b * (1. - s) + a_u * s


def sg_product(ax1, lam1, mu1, ax2, lam2, mu2):
    """Product of two SGs assuming lam1 << lam2 (sg_render.py:141-158)."""
    ratio = lam1 / lam2
    dot = torch.sum(ax1 * ax2, dim=-1, keepdim=True)
    tmp = torch.sqrt(ratio * ratio + 1. + 2. * ratio * dot)
    tmp = torch.min(tmp, ratio + 1.)
    lam3 = lam2 * tmp
    ax = (ratio / tmp) * ax1 + (1. / tmp) * ax2
    mu = mu1 * mu2 * torch.exp(lam2 * (tmp - ratio - 1.))
    return ax, lam3, mu


MU_COS, LAMBDA_COS, ALPHA_COS = 32.7080, 0.0315, 31.7003


def _cosine_integral(n, ax, lam, mu):
    axp, lamp, mup = sg_product(n, LAMBDA_COS, MU_COS, ax, lam, mu)
    d1 = torch.sum(axp * n, dim=-1, keepdim=True)
    d2 = torch.sum(ax * n, dim=-1, keepdim=True)
    return mup * hemi_integral(lamp, d1) - mu * ALPHA_COS * hemi_integral(lam, d2)


def sg_closed_form(lgt, spec, rough, albedo, normal, view):
    """render_with_sg for K=1: lgt [M,7], spec [1,3], rough [1,1], albedo/normal/view [N,3]."""
    assert spec.shape[0] == 1 and rough.shape[0] == 1
    N, M = normal.shape[0], lgt.shape[0]
    l_ax, l_lam, l_mu = (x.unsqueeze(0).expand(N, M, -1) for x in unpack_light(lgt))
    n = normal.unsqueeze(1).expand(N, M, 3)
    v = view.unsqueeze(1).expand(N, M, 3)
    r4i = 1. / (rough * rough * rough * rough)                 # [1,1]
    b_lam = (2. * r4i).reshape(1, 1, 1)
    b_mu = (r4i / PI).reshape(1, 1, 1)
    vn = torch.clamp(torch.sum(n * v, dim=-1, keepdim=True), min=0.)
    w_ax = 2 * vn * n - v
    w_ax = w_ax / (torch.norm(w_ax, dim=-1, keepdim=True) + TINY)
    w_lam = b_lam / (4 * vn + TINY)
    half = w_ax + v
    half = half / (torch.norm(half, dim=-1, keepdim=True) + TINY)
    vh = torch.clamp(torch.sum(v * half, dim=-1, keepdim=True), min=0.)
    s = spec.reshape(1, 1, 3)
    fres = s + (1. - s) * torch.pow(2.0, -(5.55473 * vh + 6.8316) * vh)
    d1 = torch.clamp(torch.sum(w_ax * n, dim=-1, keepdim=True), min=0.)
    d2 = torch.clamp(torch.sum(v * n, dim=-1, keepdim=True), min=0.)
    k = ((rough + 1.) * (rough + 1.) / 8.).reshape(1, 1, 1)
    g = (d1 / (d1 * (1 - k) + k + TINY)) * (d2 / (d2 * (1 - k) + k + TINY))
    w_mu = b_mu * (fres * g / (4 * d1 * d2 + TINY))
    f_ax, f_lam, f_mu = sg_product(l_ax, l_lam, l_mu, w_ax, w_lam, w_mu)
    spec_rgb = torch.clamp(_cosine_integral(n, f_ax, f_lam, f_mu).sum(dim=1), min=0.)
    diff = (albedo / PI).unsqueeze(1)
    diff_rgb = torch.clamp(_cosine_integral(n, l_ax, l_lam, l_mu * diff).sum(dim=1), min=0.)
    return {'sg_rgb': spec_rgb + diff_rgb, 'sg_specular_rgb': spec_rgb,
            'sg_diffuse_rgb': diff_rgb, 'sg_diffuse_albedo': albedo}


def env_radiance(lgt, dirs):
    """Background colour sum_m |mu_m| exp(|lam_m| (d . xi_m - 1)) for miss rays [N,3].

    IDRNetwork.get_background_rgb (:646-663); lobe axes normalised with +1e-8 there."""
    ax = lgt[..., :3] / (torch.norm(lgt[..., :3], dim=-1, keepdim=True) + 1e-8)
    lam = torch.abs(lgt[..., 3:4])
    mu = torch.abs(lgt[..., -3:])
    dots = torch.sum(dirs.unsqueeze(1) * ax.unsqueeze(0), dim=-1, keepdim=True)
    return (mu.unsqueeze(0) * torch.exp(lam.unsqueeze(0) * (dots - 1))).sum(1)


def light_radiance(lgt, wi):
    """Same as env_radiance but with the 1e-6 normalisation used inside the MC renderer (:1412-1418)."""
    ax, lam, mu = unpack_light(lgt)
    dots = torch.sum(wi.unsqueeze(1) * ax.unsqueeze(0), dim=-1, keepdim=True)
    return (mu.unsqueeze(0) * torch.exp(lam.unsqueeze(0) * (dots - 1))).sum(1)


# ----------------------------------------------------------------------------------------------
# samplers (path_tracing_render.py:12-271)
# ----------------------------------------------------------------------------------------------
def to_world(local, n):
    """Rotate local (z = normal) coordinates into world space (:12-33)."""
    xa = torch.zeros_like(n)
    xa[..., 0] = 1
    ya = torch.zeros_like(n)
    ya[..., 1] = 1
    up = torch.where((n[..., 0:1] > 0.9).expand(n.shape), ya, xa)
    t = torch.cross(up, n, dim=-1)
    t = t / (torch.norm(t, dim=-1, keepdim=True) + TINY)
    s = torch.cross(t, n, dim=-1)
    return local[..., :1] * t + local[..., 1:2] * s + local[..., 2:] * n


def _polar(theta, phi):
    return torch.cat([theta.sin() * phi.cos(), theta.sin() * phi.sin(), theta.cos()], dim=-1)


def pdf_cos(wi, n):
    return torch.clamp(torch.sum(wi * n, dim=-1, keepdim=True), min=TINY) / PI


def pdf_ggx(wi, n, v, rough):
    h = wi + v
    h = h / torch.norm(h, dim=-1, keepdim=True)
    bad = torch.isnan(h)
    h = torch.where(bad, n, h)
    c = torch.clamp(torch.sum(h * n, dim=-1, keepdim=True), min=TINY)
    root = c ** 2 + (1 - c ** 2) / (rough ** 4)
    pdf_h = c / (PI * (rough ** 4) * root * root)
    hv = torch.clamp(torch.sum(h * v, dim=-1, keepdim=True), min=TINY)
    return pdf_h / (4 * hv)


def _mix_weights(n, lgt):
    ax, lam, mu = unpack_light(lgt)
    e = mu.sum(dim=-1, keepdim=True)                                        # [M,1]
    nd = torch.sum(n.unsqueeze(1) * ax.unsqueeze(0), dim=-1, keepdim=True)  # [N,M,1]
    w = e.unsqueeze(0) * torch.clamp(nd, TINY)
    return w / w.sum(dim=1, keepdim=True), ax, lam, e


def pdf_mix(wi, n, lgt):
    alpha, ax, lam, _ = _mix_weights(n, lgt)
    c = lam / (2 * PI * (1 - torch.exp(-2.0 * lam)))
    dots = torch.sum(wi.unsqueeze(1) * ax.unsqueeze(0), dim=-1, keepdim=True)
    return (alpha * c.unsqueeze(0) * torch.exp(lam.unsqueeze(0) * (dots - 1))).sum(dim=1)


def sample_cos(n, r1, r2):
    theta = torch.arccos(torch.sqrt(1 - r1))
    wi = to_world(_polar(theta, 2 * PI * r2), n)
    return wi, theta.cos() / PI


def sample_ggx(n, rough, v, r1, r2):
    theta = torch.arctan(rough ** 2 * torch.sqrt(r1 / (1 - r1)))
    h = to_world(_polar(theta, 2 * PI * r2), n)
    wi = 2 * (torch.sum(v * h, dim=-1, keepdim=True)) * h - v
    return wi, pdf_ggx(wi, n, v, rough)


def sample_mix(n, lgt, r0, r1, r2):
    alpha, ax, lam, e = _mix_weights(n, lgt)
    right = torch.cumsum(alpha, dim=1)
    left = right - alpha
    right[:, -1, :] = 1.0
    left[:, 0, :] = 0.0
    cond = (r0.reshape(-1, 1, 1) >= left) & (r0.reshape(-1, 1, 1) < right)       # [N,M,1]
    k = torch.max(cond.float(), dim=1)[1].reshape(-1)                             # chosen lobe
    ax_k = ax[k]
    lam_k = lam[k]
    c_k = lam_k / (2 * PI * (1 - torch.exp(-2 * lam_k)))
    theta = torch.arccos(1.0 / lam_k * torch.log(torch.clamp(1 - lam_k * r1 / (2 * PI * c_k), TINY)) + 1)
    wi = to_world(_polar(theta, 2 * PI * r2), ax_k)
    return wi, pdf_mix(wi, n, lgt)


def draw_mis_directions(lgt, rough, normal, view, uniforms=None):
    """Three directions per point + the 3x3 pdf matrix (:1290-1325).

    ``uniforms``: [N,7] = (cos r1,r2 | ggx r1,r2 | mix r0,r1,r2); drawn from torch's global
    RNG in exactly the reference's call order when None."""
    N = normal.shape[0]
    if uniforms is None:
        u = [torch.rand(N, 1) for _ in range(4)]
        u.append(torch.rand(N, 1, 1).reshape(N, 1))
        u += [torch.rand(N, 1) for _ in range(2)]
        uniforms = torch.cat(u, dim=1)
    u = [uniforms[:, i:i + 1] for i in range(7)]
    with torch.no_grad():
        w0, p0 = sample_cos(normal, u[0], u[1])
        w1, p1 = sample_ggx(normal, rough, view, u[2], u[3])
        w2, p2 = sample_mix(normal, lgt, u[4], u[5], u[6])
        own = [torch.clamp(p, min=TINY) for p in (p0, p1, p2)]
        ws = [w0, w1, w2]
        fns = [lambda w: pdf_cos(w, normal), lambda w: pdf_ggx(w, normal, view, rough),
               lambda w: pdf_mix(w, normal, lgt)]
        table = [[own[i] if j == i else fns[j](ws[i]) for j in range(3)] for i in range(3)]
    return ws, own, table, uniforms


def mc_shade(lgt, spec, rough, albedo, normal, view, ws, own_pdf, pdf_table, visibility, indirect):
    """Sum over the 3 MIS samples of direct*vis + (1-vis)*indirect through GGX + Lambert (:1406-1476).

    spec [1,3]; rough/albedo/normal/view per point; visibility/indirect: lists of [N,1]/[N,3]."""
    spec_rgb = 0
    diff_rgb = 0
    for i in range(3):
        wi = ws[i]
        light = light_radiance(lgt, wi)
        h = wi + view
        h = h / (torch.norm(h, dim=-1, keepdim=True) + TINY)
        nh = torch.clamp(torch.sum(normal * h, dim=-1, keepdim=True), min=0)
        a2 = rough ** 2
        root = nh ** 2 + (1 - nh ** 2) / (a2 ** 2)
        D = 1.0 / (PI * (a2 ** 2) * root * root)
        vh = torch.clamp(torch.sum(view * h, dim=-1, keepdim=True), min=0.)
        fres = spec + (1. - spec) * torch.pow(2.0, -(5.55473 * vh + 6.8316) * vh)
        d1 = torch.clamp(torch.sum(view * normal, dim=-1, keepdim=True), min=0.)
        d2 = torch.clamp(torch.sum(wi * normal, dim=-1, keepdim=True), min=0.)
        k = (rough + 1.) * (rough + 1.) / 8.
        g = (d1 / (d1 * (1 - k) + k + TINY)) * (d2 / (d2 * (1 - k) + k + TINY))
        fs = fres * D * g / (4 * d1 * d2 + TINY)
        num = own_pdf[i] ** 2
        den = 0
        for j in range(3):
            den = den + pdf_table[i][j] ** 2
        weight = num / torch.clamp(den, min=TINY)
        l_all = light * visibility[i] + (1 - visibility[i]) * indirect[i]
        cosn = torch.clamp(torch.sum(wi * normal, dim=-1, keepdim=True), min=0)
        spec_rgb = spec_rgb + torch.clamp(weight * l_all * fs * cosn / own_pdf[i], min=0.)
        diff_rgb = diff_rgb + torch.clamp(weight * l_all * (albedo / PI) * cosn / own_pdf[i], min=0.)
    return {'sg_rgb': spec_rgb + diff_rgb, 'sg_specular_rgb': spec_rgb,
            'sg_diffuse_rgb': diff_rgb, 'sg_diffuse_albedo': albedo}
