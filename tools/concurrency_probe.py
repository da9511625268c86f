"""Does a kernel compute the same result when other kernels run beside it on another stream?

X (the victim) runs REPS times on stream A while Y (the load) runs back to back on stream B; every X output is compared
bit for bit with X's result on an idle chip.  Round 3's hunt for config 3's non-finite steps under trace prefetch
(tools/nan_hunt.py) ended at nefii_mis_sample producing wrong directions from correct inputs only while tracer kernels were
in flight: this probe separates the victim / load pairs.

    python tools/concurrency_probe.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nefii_amd import conf, ops, synthetic as syn
from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork

dev = 'cuda'
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def main():
    mc = syn.model_conf('conf')
    sd = syn.make_state_dict(mc, seed=0, scene='bowl')
    model = IDRNetwork(conf.from_dict(mc))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    model.freeze_geometry()
    net = model.implicit_network
    pm = net.packed(f16x3=True)
    g = torch.Generator().manual_seed(3)
    n = 114891
    normal = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    view = torch.nn.functional.normalize(torch.randn(n, 3, generator=g) * 0.2 + torch.tensor([0., 0., 1.]), dim=-1).to(dev)
    rough = (torch.rand(n, 1, generator=g) * 0.8 + 0.15).to(dev)
    uni = torch.rand(n, 7, generator=g).to(dev)
    lgt = model.envmap_material_network.get_lgtSGs().detach().clone()
    xs = (torch.randn(1 << 19, 3, generator=g) * 0.45).to(dev)
    xsmall = (torch.randn(20000, 3, generator=g) * 0.45).to(dev)
    big = torch.randn(1 << 22, generator=g).to(dev)
    a_mat = torch.randn(2048, 2048, generator=g).to(dev)
    rad = model.rendering_network
    feat = (torch.randn(60000, mc['feature_vector_size'], generator=g) * 0.3).to(dev)
    p60 = (torch.randn(60000, 3, generator=g) * 0.4).to(dev)
    n60 = torch.nn.functional.normalize(torch.randn(60000, 3, generator=g), dim=-1).to(dev)

    def radiance():
        with torch.no_grad():
            return rad(p60, n60, n60, feat)

    victims = {
        'mis_sample': lambda: torch.cat([t.reshape(-1) for t in ops.mis_sample(lgt, rough, normal, view, uni)]),
        'env_radiance': lambda: ops.EnvRadianceFn.apply(lgt, normal, 1e-6).reshape(-1),
        'torch sin*cos+sqrt (elementwise)': lambda: torch.sin(big) * torch.cos(big) + torch.sqrt(big.abs()),
        'torch where/compare chain': lambda: torch.where(big > 0.3, big * 2, torch.where(big < -0.3, -big, big * big)),
        'sdf_eval split (20k pts)': lambda: ops.sdf_eval(pm, xsmall),
        'sdf_eval coarse (20k pts)': lambda: ops.sdf_eval(pm, xsmall, coarse=True),
        'sdf_value_grad (20k pts)': lambda: torch.cat([t.reshape(-1) for t in ops.sdf_value_grad(pm, xsmall, want_feat=True)]),
        'radiance forward (60k pts)': lambda: radiance().reshape(-1),
    }
    loads = {
        'idle': None,
        'sdf_eval split (512k pts)': lambda: ops.sdf_eval(pm, xs),
        'sdf_eval coarse (512k pts)': lambda: ops.sdf_eval(pm, xs, coarse=True),
        'sdf_value_grad (512k pts)': lambda: ops.sdf_value_grad(pm, xs, want_feat=True),
        'torch matmul 2048^3': lambda: a_mat @ a_mat,
        'torch elementwise': lambda: torch.sin(big) * torch.cos(big),
    }
    if os.environ.get('PROBE_LOADS', '') == 'wide':
        # which of the MFMA kernels of this library disturb a packed-fp32 victim (mis_sample is the canary)?
        victims = {k: v for k, v in victims.items() if k in ('mis_sample', 'env_radiance')}
        mc2 = syn.model_conf('neus')
        m2 = IDRNetwork(conf.from_dict(mc2))
        m2.load_state_dict(syn.make_state_dict(mc2, seed=0, scene='bowl'), strict=True)
        m2 = m2.to(dev)
        m2.freeze_geometry()
        pm2 = m2.implicit_network.packed(f16x3=True)
        mat = model.envmap_material_network
        feat_b = (torch.randn(200000, mc['feature_vector_size'], generator=g) * 0.3).to(dev)
        p_b = (torch.randn(200000, 3, generator=g) * 0.4).to(dev)
        n_b = torch.nn.functional.normalize(torch.randn(200000, 3, generator=g), dim=-1).to(dev)
        h_a = torch.randn(4096, 4096, generator=g).to(dev).half()

        def rad_train():
            out = rad(p_b, n_b, n_b, feat_b)
            out.sum().backward()

        def mat_fwd():
            with torch.no_grad():
                return mat(p_b, feat_b, n_b)

        loads = {
            'idle': None,
            'conf coarse 16s<4,4> (512k)': lambda: ops.sdf_eval(pm, xs, coarse=True),
            'conf split 16q<4,4> (512k)': lambda: ops.sdf_eval(pm, xs),
            'neus coarse 16s<6,2> (512k)': lambda: ops.sdf_eval(pm2, xs, coarse=True),
            'neus split 16q<2> (512k)': lambda: ops.sdf_eval(pm2, xs),
            'neus value_grad (512k)': lambda: ops.sdf_value_grad(pm2, xs, want_feat=False),
            'radiance fwd+bwd+wgrad (200k)': rad_train,
            'material fwd (200k)': mat_fwd,
            'torch half matmul 4096^3': lambda: h_a @ h_a,
        }
    if os.environ.get('PROBE_LOADS', '') == 'torch':
        # are torch's own kernels (the step's glue: weight-norm, normalisations, Adam's arithmetic) disturbed beside the
        # single-pass evaluator?  (run with a library built WITHOUT the register claim: NEFII_LIB_PATH)
        b1, b2, b3 = (torch.randn(1 << 21, generator=g).to(dev) for _ in range(3))
        v_w = torch.randn(512, 512, generator=g).to(dev)
        g_w = torch.rand(512, 1, generator=g).to(dev) + 0.5
        x3 = torch.randn(200000, 3, generator=g).to(dev)
        params = [torch.randn(512, 512, generator=g).to(dev) for _ in range(12)]
        grads = [torch.randn(512, 512, generator=g).to(dev) * 1e-3 for _ in range(12)]

        def adam_like():
            m = torch._foreach_lerp([torch.zeros_like(p_) for p_ in params], grads, 0.1)
            vv = torch._foreach_addcmul([torch.zeros_like(p_) for p_ in params], grads, grads, 0.001)
            den = torch._foreach_sqrt(vv)
            torch._foreach_add_(den, 1e-8)
            out = torch._foreach_addcdiv(params, m, den, -5e-4)
            return torch.cat([o.reshape(-1) for o in out])

        def fused_adam():
            ps = [p_.clone().requires_grad_(True) for p_ in params]
            for p_, g_ in zip(ps, grads):
                p_.grad = g_.clone()
            opt = torch.optim.Adam(ps, lr=5e-4, fused=True)
            opt.step()
            opt.step()
            return torch.cat([p_.detach().reshape(-1) for p_ in ps])

        victims = {
            'addcmul': lambda: torch.addcmul(b1, b2, b3, value=0.5),
            'lerp': lambda: torch.lerp(b1, b2, 0.3),
            'a*b+c*a (fused by nothing: 3 kernels)': lambda: b1 * b2 + b3 * b1,
            'normalize rows [n,3]': lambda: x3 / (x3.norm(dim=-1, keepdim=True) + 1e-6),
            'weight_norm 512x512': lambda: torch._weight_norm(v_w, g_w, 0),
            'sigmoid * (1-0.089) + 0.089': lambda: torch.sigmoid(b1) * (1 - 0.089) + 0.089,
            'foreach Adam arithmetic': adam_like,
            'torch.optim.Adam(fused=True) x2': fused_adam,
            'mis_sample (this library)': victims['mis_sample'],
        }
        loads = {'idle': None, 'conf coarse 16s<4,4> (512k)': lambda: ops.sdf_eval(pm, xs, coarse=True),
                 'conf split 16q<4,4> (512k)': lambda: ops.sdf_eval(pm, xs)}
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    print('victim x load: runs that differ from the idle-chip result / runs   (elements that differ in the worst run)')
    for vname, vf in victims.items():
        ref = vf()
        torch.cuda.synchronize()
        again = vf()
        torch.cuda.synchronize()
        det = torch.equal(torch.nan_to_num(ref), torch.nan_to_num(again))
        row = []
        for lname, lf in loads.items():
            outs = []
            torch.cuda.synchronize()
            if lf is not None:
                with torch.cuda.stream(sb):
                    for _ in range(max(6, REPS // 3)):
                        lf()
            with torch.cuda.stream(sa):
                for _ in range(REPS):
                    outs.append(vf())
            torch.cuda.synchronize()
            bad = [int((torch.nan_to_num(o) != torch.nan_to_num(ref)).sum()) for o in outs]
            row.append('%s: %d/%d (%d)' % (lname, sum(b > 0 for b in bad), len(bad), max(bad)))
        print('%-34s deterministic alone: %s | %s' % (vname, det, ' | '.join(row)), flush=True)


if __name__ == '__main__':
    main()
