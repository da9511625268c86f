"""CPU emulation of candidate arithmetics for the SDF network (VERDICT r4 next #1b / #1c): what would they do to the values the
path consumes, before anyone writes a kernel for them?

  split      today's split-precision evaluator: x_h w_h + x_h w_l + x_l w_h on fp16 hi/lo pairs (x 16, x 64), fp32 accumulate
  single     the coarse evaluator: x_h w_h
  2prod_x    x_h w_h + x_l w_h  (activations split, weights rounded to fp16)
  2prod_w    x_h w_h + x_h w_l  (weights split, activations rounded)
  fp8corr    x_h w_h + x_h q(w_l) + q(x_l) w_h, q = e4m3 with one power-of-two scale per 32 consecutive K (the block-scaled
             v_mfma_scale_f32_16x16x128_f8f6f4: 3 x the fp16 form's sustained rate on this part, profiles/r05/fp8_probe.txt)
  fp8corr_fix the same with ONE constant power-of-two scale per operand kind instead of block scales (what "16f" does)
  fp8corr_hw x_h w_h + q(x_h) q(w_l) + q(x_l) q(w_h): the same with BOTH operands of each correction product in e4m3 - the
             instruction has no fp16 x fp8 form, so this is what a kernel can actually compute (round 6)

(1) the SDF value on points of the bounding sphere and within 0.02 of the surface: max / rms error against fp64 - to be read
    against the tracer's decision threshold 5e-5, the coarse bound tau ~ 2e-3 and the split evaluator's 5e-7;
(2) #1c, features and normals at hit points (the sdf_value_grad pass: forward AND input-backward in the candidate arithmetic)
    pushed through the oracle's material / radiance networks and MC shading on config 3's sample: relative L2 of rendered RGB
    and albedo against the same oracle with fp32 features - the north-star bound is 1e-3 and the HIP path sits at 1e-5 today.
Emulation: operands rounded as the kernels round them, products summed in fp64 and rounded to fp32 once per layer (the
kernels' fp32 accumulation order adds ~1e-7 on top; irrelevant at the scales compared).  CPU only; prints a table."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from nefii_amd import synthetic as syn
from oracle import nets, renderer as orr

MODES = ['split', 'single', '2prod_x', '2prod_w', 'fp8corr', 'fp8corr_hw', 'fp8corr_fix']


def e4m3(v):
    """round to the nearest OCP e4m3 value (|v| <= 448 assumed), fp64 in / out"""
    a = v.abs()
    e = torch.floor(torch.log2(a.clamp_min(2.0 ** -20))).clamp_min(-6.0)        # normal exponents -6..8; below: subnormal step
    step = torch.pow(2.0, e - 3.0)
    return torch.sign(v) * torch.round(a / step) * step


def q8(v, e):
    """e4m3 image of v x 2^e with saturation at +-448 (one CONSTANT scale for the whole tensor), back in v's units"""
    return e4m3((v * 2.0 ** e).clamp(-448.0, 448.0)) * 2.0 ** -e


def mx8(v):
    """block-scaled e4m3 along the last axis: one power-of-two scale per 32 elements"""
    K = v.shape[-1]
    pad = (-K) % 32
    if pad:
        v = torch.nn.functional.pad(v, (0, pad))
    b = v.reshape(*v.shape[:-1], -1, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -60)
    scale = torch.pow(2.0, torch.ceil(torch.log2(amax / 448.0)))
    q = (e4m3(b / scale) * scale).reshape(*v.shape)
    return q[..., :K]


def emu_matmul(x, w, mode):
    """x [N,K] @ w[M,K]^T in the candidate arithmetic (fp32 in / out)"""
    if mode == 'f32':
        return x @ w.t()
    xs, ws = x.double() * 16.0, w.double() * 64.0
    xh = xs.half().double()
    xl = (xs - xh).half().double()
    wh = ws.half().double()
    wl = (ws - wh).half().double()
    acc = xh @ wh.t()
    if mode == 'split':
        acc = acc + xh @ wl.t() + xl @ wh.t()
    elif mode == '2prod_x':
        acc = acc + xl @ wh.t()
    elif mode == '2prod_w':
        acc = acc + xh @ wl.t()
    elif mode == 'fp8corr':
        acc = acc + xh @ mx8(wl).t() + mx8(xl) @ wh.t()
    elif mode == 'fp8corr_hw':
        # what the hardware can do (round 6): v_mfma_scale_f32_16x16x128_f8f6f4 takes BOTH operands in an f8f6f4 format - the
        # partner of each low part is the e4m3 image of the high part, not the high part itself
        acc = acc + mx8(xh) @ mx8(wl).t() + mx8(xl) @ mx8(wh).t()
    elif mode == 'fp8corr_fix':
        # the evaluator as built in round 6 ("16f"): no data-dependent block scales - constants (x_h 2^-3, w_l 2^10, x_l 2^8,
        # w_h 2^-1 on the 16- / 64-scaled operands), restored by the instruction's scale operand
        acc = acc + q8(xh, -3) @ q8(wl, 10).t() + q8(xl, 8) @ q8(wh, -1).t()
    return (acc / 1024.0).float()


class EmuLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, mode):
        ctx.save_for_backward(w)
        ctx.mode = mode
        return emu_matmul(x, w, mode) + b

    @staticmethod
    def backward(ctx, g):
        w, = ctx.saved_tensors
        return emu_matmul(g.contiguous(), w.t().contiguous(), ctx.mode), None, None, None


def sdf_forward_emu(sd, cfg, x, mode, prefix='implicit_network'):
    skip_in = tuple(cfg.get('skip_in', ()))
    enc = nets.posenc(x, int(cfg.get('multires', 0)))
    n_lin = nets.count_layers(sd, prefix)
    h, feat = enc, None
    for l in range(n_lin):
        if cfg.get('use_last_as_f') and l == n_lin - 1:
            feat = h
        if l in skip_in:
            h = torch.cat([h, enc], dim=1) / math.sqrt(2)
        w, b = nets.linear_params(sd, '%s.lin%d' % (prefix, l))
        h = EmuLinear.apply(h, w, b, mode)
        if l < n_lin - 1:
            h = torch.nn.functional.softplus(h, beta=100)
    return torch.cat([h, feat], dim=-1) if cfg.get('use_last_as_f') else h


def main():
    torch.set_num_threads(8)
    w = syn.WORKLOADS['cfg3']
    mc, sd = syn.workload_state_dict('cfg3', seed=0)
    cfg = mc['implicit_network']
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6000, 3, generator=g)
    x = x / x.norm(dim=1, keepdim=True) * torch.rand(6000, 1, generator=g) ** (1 / 3.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        ref = nets.sdf_forward(sd64, cfg, x.double())[:, 0]
        near = ref.abs() < 0.02
        print('SDF value against fp64, %d points of the unit ball (%d within 0.02 of the surface)' % (x.shape[0], int(near.sum())))
        for mode in MODES:
            v = sdf_forward_emu(sd, cfg, x, mode)[:, 0].double()
            e = (v - ref).abs()
            print('  %-8s max %.2e  rms %.2e | near the surface: max %.2e  rms %.2e' % (
                mode, e.max(), e.pow(2).mean().sqrt(), e[near].max(), e[near].pow(2).mean().sqrt()))
    if len(sys.argv) > 1 and sys.argv[1] == 'values':
        return
    # ---- (2) features / normals at hit points in the candidate arithmetic -> rendered RGB / albedo (oracle, config 3 sample)
    inp, gt = syn.make_inputs(64, w['image_hw'], w['focal'], w['cam_pos'], 16, seed=1)
    B, S, R, _ = inp['uv'].shape
    flat = dict(inp)
    flat['uv'] = inp['uv'].reshape(B, S * R, 2)
    flat['object_mask'] = inp['object_mask'].reshape(B, S, 1).expand(B, S, R).reshape(B, S * R)
    n_ray = S * R
    steps1, steps2 = torch.rand(100, generator=g), torch.rand(100, generator=g)
    uniforms = torch.rand(n_ray, 7, generator=g)
    class EmuRenderer(orr.Renderer):
        mode = 'f32'

        def surface_terms(self, pts):       # get_rbg_value's implicit_network(x) + gradient(x) in the candidate arithmetic
            if self.mode == 'f32':
                return super().surface_terms(pts)
            with torch.enable_grad():
                xr = pts.detach().clone().requires_grad_(True)
                y = sdf_forward_emu(self.sd, self.sdf_cfg, xr, self.mode)
                g_, = torch.autograd.grad(y[:, :1].sum(), xr)
            feats = y[:, 1:].detach() if self.F > 0 else None
            return feats, orr._unit(g_)

    outs = {}
    for mode in ['f32'] + MODES:
        R_ = EmuRenderer(sd, mc, training=True)
        R_.dead_work = False
        R_.mode = mode
        with torch.no_grad():
            out = R_.forward(flat, steps1, uniforms, steps2)
        outs[mode] = {k: out[k].detach() for k in ('sg_rgb_values', 'sg_diffuse_albedo_values', 'normal_values', 'network_object_mask',
                                                   'secondary_dir', 'secondary_mask', 'sg_roughness_values')}
        outs[mode]['_ray_hit'] = out['_ray_hit']
    base = outs['f32']
    m = base['network_object_mask']
    print('features / normals at hit points in the candidate arithmetic (value, features: forward; normals: forward AND input-'
          'backward emulated), %d rays, %d hit; relative L2 against the same oracle with the fp32 surface pass:' % (n_ray, int(m.sum())))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from parity import mc_flagged_rays
    for mode in MODES:
        o = outs[mode]
        # rays whose Monte-Carlo sampling differs DISCRETELY (another lobe picked at a CDF boundary, a secondary ray hitting on
        # one side only) are counted and set apart, as the GPU suite does (tests/parity.py): their colour is another sample
        flagged, n_dir, n_vis = mc_flagged_rays(o, base, o['_ray_hit'], base['_ray_hit'])
        keep = m & ~flagged
        rl = lambda k, sel: ((o[k][sel] - base[k][sel]).norm() / base[k][sel].norm()).item()
        print('  %-8s rgb %.2e (all hit rays: %.2e; %d rays with another sampled direction, %d with another secondary hit flag)  '
              'albedo %.2e  roughness %.2e  normals %.2e' % (mode, rl('sg_rgb_values', keep), rl('sg_rgb_values', m), n_dir, n_vis,
                                                              rl('sg_diffuse_albedo_values', m), rl('sg_roughness_values', m),
                                                              rl('normal_values', m)))


if __name__ == '__main__':
    main()
