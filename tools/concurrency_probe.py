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
    if os.environ.get('PROBE_LOADS', '') == 'forms':
        # WHICH packed-fp32 instruction forms are victims?  (libnefii_canary.so: tests/canary/pk_forms.hip, one form per
        # kernel in inline assembly, launch shape of mis_sample; run with NEFII_LIB_PATH = a build WITHOUT the register claim)
        import ctypes
        from nefii_amd import build
        from nefii_amd.ops import _ptr
        build.build_canary(verbose=False)
        can = ctypes.CDLL(build.CANARY_OUT)
        P, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        can.nefii_canary_pk_form.restype = I
        can.nefii_canary_pk_form.argtypes = [I, P, P, I64, I, P]
        can.nefii_mis_sample.restype = I
        can.nefii_mis_sample.argtypes = [P, I, P, P, P, P, I64, P, P, P, P]
        vin = torch.empty(n, 6)
        vin[:, 0:2] = 0.9 + 0.09 * torch.rand(n, 2, generator=g)        # |a| < 1
        vin[:, 2:4] = 0.1 * torch.randn(n, 2, generator=g)
        vin[:, 4:6] = torch.randn(n, 2, generator=g)
        vin = vin.to(dev)
        iters = int(os.environ.get('PROBE_ITERS', '400'))

        def form(fi):
            def run():
                out = torch.empty(n, 2, device=dev)
                assert can.nefii_canary_pk_form(fi, _ptr(vin), _ptr(out), n, iters, torch.cuda.current_stream().cuda_stream) == 0
                return out.reshape(-1)
            return run

        def mis_pk():
            wi = torch.empty(3, n, 3, device=dev)
            own = torch.empty(3, n, device=dev)
            tab = torch.empty(3, n, 3, device=dev)
            r = rough.reshape(-1).contiguous()
            assert can.nefii_mis_sample(_ptr(lgt), lgt.shape[0], _ptr(r), _ptr(normal), _ptr(view), _ptr(uni), n, _ptr(wi),
                                        _ptr(own), _ptr(tab), torch.cuda.current_stream().cuda_stream) == 0
            return torch.cat([wi.reshape(-1), own.reshape(-1), tab.reshape(-1)])
        names = ['v_pk_fma_f32 (plain)', 'v_pk_fma_f32 op_sel_hi:[1,0,1] (broadcast)', 'v_pk_fma_f32 neg_lo/neg_hi',
                 'v_pk_mul_f32 op_sel (swapped halves) + v_pk_add_f32', 'v_pk_mul_f32 neg + v_pk_add_f32 neg',
                 'v_pk_mov_b32 op_sel + v_pk_fma_f32', 'scalar v_fma_f32 x 2 (control)',
                 'compiler-formed: tangent frame + change of basis, rsq', 'compiler-formed: cross products, mul/add only',
                 'compiler-formed: sin/cos directions + dots']
        tnames = ['v_rsq_f32; v_pk_mul_f32 OVERWRITES the pair holding its source', 'v_rsq_f32; v_mul_f32 overwrites its source (control)',
                  'v_rsq_f32; s_nop 0; v_pk_mul_f32 consumes the result', 'v_rsq_f32; v_pk_mul_f32 consumes the result (no wait state)',
                  'v_rsq_f32; independent v_pk_fma_f32; v_pk_mul_f32 consumes', 'v_rsq_f32; s_nop 0; v_mul_f32 consumes (control)',
                  'v_rsq_f32 v42; v_pk_mov_b32 v[42:43]; v_rsq_f32 v42',
                  'cos, sin, sin, cos; v_pk_mul_f32 consumes three of the results', 'cos, sin, sin, cos; 2 x v_mul_f32 consume (control)',
                  'cos, sin, sin, cos; 3 plain instructions; v_pk_mul_f32 consumes', 'cos, sin, sin, cos; v_pk_mul_f32 consumes the last pair']
        victims = {}
        which = os.environ.get('PROBE_FORMS', 'all')
        for claim in (0, 16):
            for i in range(10):
                if which == 'all' or (which == 'trans' and claim == 0 and i in (7, 8, 9)):
                    victims[names[i] + (' [74 VGPRs]' if claim else '')] = form(i + claim)
        if which != 'opsel':
            for i in range(len(tnames)):
                victims[tnames[i]] = form(32 + i)
        onames = ['v_pk_mul_f32 op_sel:[0,1] (src1 high half to both lanes) + v_pk_add_f32', 'v_pk_mul_f32 op_sel:[1,0] (src0 high half to both lanes) + v_pk_add_f32',
                  'v_pk_fma_f32 op_sel:[0,1,0]', 'v_pk_mul_f32 + v_pk_add_f32 op_sel:[0,1]', 'v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 halves swapped) + v_pk_add_f32',
                  'v_pk_mul_f32 op_sel:[0,1], src1 = the running value']
        for i in range(6):
            victims[onames[i]] = form(10 + i)
        victims['mis_sample compiled WITH packed fp32 (positive control)'] = mis_pk
        bigm = big.reshape(4096, -1)
        victims['torch.linalg.vector_norm(4096 x 1024 fp32, dim=1) (reduce_kernel<NormOps>: has the form)'] = lambda: torch.linalg.vector_norm(bigm, dim=1)
        victims['torch.linalg.vector_norm(4096 x 1024 fp32, dim=0)'] = lambda: torch.linalg.vector_norm(bigm, dim=0)
        victims['torch.linalg.vector_norm(4M fp32)'] = lambda: torch.linalg.vector_norm(big).reshape(1)
        victims['torch.logaddexp2 (4M fp32) (has the form)'] = lambda: torch.logaddexp2(big, big * 0.5)
        victims['mis_sample of the library under test (compiled WITHOUT packed fp32)'] = lambda: torch.cat(
            [t.reshape(-1) for t in ops.mis_sample(lgt, rough, normal, view, uni)])
        lgt = lgt.contiguous()
        loads = {'idle': None, 'conf coarse 16s<4,4> (512k)': lambda: ops.sdf_eval(pm, xs, coarse=True),
                 'conf split 16q<4,4> (512k)': lambda: ops.sdf_eval(pm, xs)}
    if os.environ.get('PROBE_LOADS', '') == 'forms' and os.environ.get('PROBE_NEIGHBOURS'):
        # WHAT in the neighbour disturbs the victim form?  Synthetic 8-wave / 216-VGPR neighbours, one instruction kind each
        can.nefii_canary_neighbour.restype = I
        can.nefii_canary_neighbour.argtypes = [I, P, I, I, P]
        sink = torch.zeros(512, device=dev)
        kinds = ['v_mfma_f32_16x16x32_f16', 'v_pk_fma_f16', 'v_pk_fma_f32', 'v_fma_f32', 'ds_read_b128', 'v_exp_f16', 's_nop only',
                 'v_mfma + v_pk_fma_f16', 'v_pk_mul_f16 v, v, s op_sel_hi:[1,0]', 'v_pk_mul_f16 v, v, v op_sel_hi:[1,0]',
                 'v_pk_fma_f16 v, v, v, s op_sel_hi:[1,1,0]', 'v_pk_max_f16 v, v, 0', 'ds_bpermute_b32', 'v_exp_f16_sdwa',
                 'v_pack_b32_f16 + v_cvt_pk_f16_f32', 'v_pk_fma_f16 v, v, s, v op_sel_hi:[1,0,0]']
        nit = [1000, 4000, 4000, 4000, 2000, 2000, 4000, 1000, 4000, 4000, 4000, 4000, 1000, 2000, 2000, 4000]

        def neighbour(k):
            def run():
                assert can.nefii_canary_neighbour(k, _ptr(sink), nit[k], 2048, torch.cuda.current_stream().cuda_stream) == 0
            return run
        loads = {'idle': None}
        for k in range(len(kinds)):
            loads['beside ' + kinds[k]] = neighbour(k)
        loads['beside the single-pass evaluator'] = lambda: ops.sdf_eval(pm, xs, coarse=True)
        victims = {k: v for k, v in victims.items() if k.startswith('v_pk_mul_f32 op_sel:[0,1] (src1') or k.startswith('v_pk_mul_f32 op_sel:[1,0]')}
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
