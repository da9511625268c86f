"""GPU parity tests on BASELINE.json's configurations 2-5: every config's own model, geometry
stand-in, camera and rays-per-pixel, (a) shrunk in pixels only so that the CPU oracle finishes in seconds - outputs,
loss, parameter gradients and SDF-evaluation counters against it - and (b) at full size through size-independent
properties.  Config 1 at full size is tests/test_gpu_renderer.py::test_train_step_full_size_vs_oracle."""
import math
import time

import pytest
import torch

from nefii_amd import conf, ops, synthetic as syn
from oracle import renderer as orr
from parity import FLOAT_KEYS, compare_outputs, mc_flagged_rays, rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def build_model(mc, sd, training=True):
    from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
    m = IDRNetwork(conf.from_dict(mc))
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    m.freeze_geometry()
    m.train(training)
    return m


def to_dev(inp):
    return {k: v.to(DEV) for k, v in inp.items()}


def per_ray_layout(inp, gt=None):
    """[1,S,R,2] multi-ray pixels -> [1,S*R,2]: every sub-pixel ray its own 'pixel' (same rays, same order), so that
    outputs can be compared ray by ray before the mean over R hides which ray differs."""
    uv = inp['uv']
    if uv.dim() == 3:
        return inp, gt, 1
    B, S, R, _ = uv.shape
    flat = dict(inp)
    flat['uv'] = uv.reshape(B, S * R, 2)
    flat['object_mask'] = inp['object_mask'].reshape(B, S, 1).expand(B, S, R).reshape(B, S * R)
    if gt is not None:
        gt = gt.reshape(B, S, 1, 3).expand(B, S, R, 3).reshape(B, S * R, 3)
    return flat, gt, R


def gpu_forward_with_per_ray_draws(m, inp, uniforms):
    """IDRNetwork.forward_with_uv in its two public halves, with the sampler's draws chosen per RAY before it is known
    which rays hit (the oracle takes the same [N_ray, 7] table)."""
    ctx = m.trace_head(inp)
    hit = ctx['network_object_mask']
    if uniforms is not None:
        m.uniforms_override = uniforms.to(DEV)[hit]
    out = m.shade_tail(ctx, torch.nonzero(hit).flatten())
    m.uniforms_override = None
    return out


SHRUNK = {'cfg2': 256, 'cfg3': 48, 'cfg4': 48}
# 'cfg2-coarse': 2048 rays - above RayTracing.coarse_min_rays, so that config 2's model-level path is ALSO compared with the oracle
# with the coarse pass and the staged min-SDF search in it (the 256-ray case runs every sample in split precision)
SHRUNK_CASE = {'cfg2-coarse': 2048, 'cfg2-coarse-tier': 2048}
# measured on these exact workloads (round 4): every count of discrete differences between the HIP path and the oracle
# (a sampled lobe that differs = a uniform draw within rounding of a CDF boundary: 3 rays of config 3's 3072 with the
# replicated embedding of the stand-in geometry, whose feature vector - and with it the lobe weights - differs from the
# zero-padded one's, where it was 0-1 by box)
PINNED_DISCRETE = {'cfg2': {'flips': 0}, 'cfg3': {'flips': 0, 'dir': 2, 'vis': 0}, 'cfg3-bowl': {'flips': 0, 'dir': 2, 'vis': 0},
                   'cfg3-dense': {'flips': 0, 'dir': 5, 'vis': 0}, 'cfg4': {'flips': 0}, 'cfg4-dense': {'flips': 0},
                   'cfg3-tier': {'flips': 0, 'dir': 20, 'vis': 0}, 'cfg4-tier': {'flips': 0},
                   'cfg3-dense-tier': {'flips': 0, 'dir': 30, 'vis': 0}, 'cfg3-frame': {'flips': 2, 'dir': 8, 'vis': 2},
                   'cfg2-coarse': {'flips': 0}, 'cfg2-coarse-tier': {'flips': 0},
                   # '-fp8': nefii_tracer_params.split_fp8 on top of the tier (round 6; the tier's class of effect, same allowances)
                   'cfg3-tier-fp8': {'flips': 0, 'dir': 30, 'vis': 1}}


@pytest.mark.parametrize('wl', ['cfg2', 'cfg3', 'cfg3-bowl', 'cfg3-dense', 'cfg4', 'cfg4-dense', 'cfg3-tier', 'cfg4-tier',
                                'cfg3-dense-tier', 'cfg3-frame', 'cfg2-coarse', 'cfg2-coarse-tier', 'cfg3-tier-fp8'])
def test_config_shrunk_in_pixels_vs_oracle(wl):
    """The config's model at full network width, its geometry stand-in, camera and rays per pixel (64 for configs 3-4);
    only the number of pixels is reduced.  Forward + IDRLoss + backward against the CPU oracle with injected draws:
    north-star tolerance on RGB / albedo ray by ray, loss terms, every parameter gradient, and the tracer's
    SDF-evaluation counters against the oracle's evaluation counts."""
    from nefii_amd.model.loss import IDRLoss
    # 'cfg3-bowl': config 3 on the ZERO-PADDED embedding of its stand-in geometry with round 3's gradient bound (3e-3): the
    # bound of the replicated embedding (6e-3) follows that embedding's feature vector, and this case keeps the kernels'
    # arithmetic (half training state included) pinned where the embedding did not move
    # '-tier': the tiered sphere tracing FORCED on (the full-size configs take it by default, these shrunk ones - 3072 rays -
    # would not): the one arithmetic of the path that changes values, held here to its parity table:
    # no hit-mask flip, hit points within 1e-4, RGB and albedo at the north-star 1e-3 (measured 5e-5 .. 2.5e-4), the other
    # channels at 4e-3 (roughness of the random-weight material net 2.1e-3), at most 30 rays with another sampled lobe
    # (measured 15), gradients at the untiered bound, evaluation counts within 1 % of the oracle's
    tier = '-tier' in wl
    fp8 = wl.endswith('-fp8')
    bound_key, wl = wl, wl.split('-')[0]
    w = syn.WORKLOADS[wl]
    # the configs' own stand-in is the network TRAINED at full width by the Step-1 runner (tools/train_scene_sdf.py;
    # nefii_amd/assets/scene_*_sdf512.npz); '-dense' / '-bowl': round 4's replicated / rounds 2-4's zero-padded embedding of the
    # 8 x 64 fit; '-frame': the network trained on the thin-feature scene
    scene = None
    for part in bound_key.split('-')[1:]:
        scene = {'bowl': 'bowl', 'dense': 'bowl_dense', 'frame': 'frame_trained'}.get(part, scene)
    mc, sd = syn.workload_state_dict(wl, seed=0, scene=scene)
    lc = syn.loss_conf(w['model'])
    # (the 12 patches of seed 1 all miss the thin frame; seed 4's see bars, plate and ball: hit fraction 0.24)
    inp, gt = syn.make_inputs(SHRUNK_CASE.get(bound_key, SHRUNK[wl]), w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'],
                              seed=4 if scene == 'frame_trained' else 1)
    flat, gt_flat, R = per_ray_layout(inp, gt)
    n_ray = flat['uv'].shape[1]
    g = torch.Generator().manual_seed(5)
    steps1, steps2 = torch.rand(100, generator=g), torch.rand(100, generator=g)
    mc_shading = mc.get('render_type', 'sg') != 'sg'
    uniforms = torch.rand(n_ray, 7, generator=g) if mc_shading else None
    # ---- oracle
    sdo = {k: v.clone() for k, v in sd.items()}
    for k in sdo:
        if not k.startswith('implicit') and not (k.endswith('specular_reflectance') and
                                                 mc['envmap_material_network'].get('fix_specular_albedo')):
            sdo[k].requires_grad_(True)
    Ro = orr.Renderer(sdo, mc, training=True)
    Ro.dead_work = False
    t0 = time.time()
    ref = Ro.forward(flat, steps1, uniforms, steps2)
    rlo = orr.idr_loss(ref, gt_flat, lc)
    rlo['loss'].backward()
    t_oracle = time.time() - t0
    # ---- HIP path
    m = build_model(mc, sd, True)
    m.secondary_miss_search = True      # as the oracle (and the reference) runs the secondary trace: its evaluation counts
    m.ray_tracer.minsdf_steps_override = [steps1, steps2] if mc_shading else steps1
    m.ray_tracer.collect_counters = True
    m.ray_tracer.counter_sum = None
    m.ray_tracer.trace_tier = tier
    m.ray_tracer.split_fp8 = fp8
    out = gpu_forward_with_per_ray_draws(m, to_dev(flat), uniforms)
    stats = compare_outputs(out, ref, max_flips=max(2, n_ray // 1000), what=bound_key + ' shrunk', rays_per_pixel=1,
                            ray_hit=m.last_ray_hit, ref_ray_hit=ref['_ray_hit'], max_explained_frac=0.02,
                            tol_aux=4e-3 if tier else None)
    if 'coarse' in bound_key:
        c = m.ray_tracer.counter_sum
        assert c[:, 5].sum() > 0 and c[:, 11].sum() > 0 and c[:, 12].max() == 0, 'the coarse pass / the staged search did not run'
    if tier:
        c9 = m.ray_tracer.counter_sum[:, 9].sum().item()
        assert c9 > 0.3 * (c9 + m.ray_tracer.counter_sum[:, 0].sum().item()), 'the tier did not run'
    print('[%s] %d rays, oracle %.1f s, hit fraction %.3f, discrete differences %s' % (
        wl, n_ray, t_oracle, ref['_ray_hit'].float().mean(), stats))
    # where the measured count of discrete differences IS zero it is asserted to be zero (the allowances above are for
    # workloads with knife-edge rays): hit-mask flips, rays with another sampled lobe, secondary rays hitting on one side only
    for k, allowed in PINNED_DISCRETE[bound_key].items():
        assert stats[k] <= allowed, (bound_key, k, stats[k], allowed)
    if mc_shading:
        sm, rsm = m.last_ray_hit.cpu(), ref['_ray_hit']
        # the indirect branch does real work here (the thin frame re-hits itself less often than the bowl)
        assert ref['secondary_mask'].float().mean().item() > (0.03 if scene == 'frame_trained' else 0.2)
        if torch.equal(sm, rsm):
            assert (out['secondary_mask'].cpu() != ref['secondary_mask']).float().mean().item() < 0.005
    # ---- loss and gradients (rays with a discrete MC difference are part of both sums: bounded above at 2 % of the rays)
    lo = IDRLoss(**lc)(out, {'rgb': gt_flat.to(DEV)})
    for k in ('loss', 'idr_rgb_loss', 'sg_rgb_loss', 'mask_loss', 'normalsmooth_loss', 'background_rgb_loss'):
        assert abs(lo[k].item() - rlo[k].item()) <= 5e-3 * abs(rlo[k].item()) + 1e-6, (k, lo[k].item(), rlo[k].item())
    lo['loss'].backward()
    worst = 0.0
    for name, p in m.named_parameters():
        gref = sdo[name].grad
        if gref is not None and gref.norm() > 0:
            assert p.grad is not None, name
            worst = max(worst, rel_l2(p.grad, gref))
            # 3 x the worst value measured in round 3 (5.1e-4 / 9.3e-4 / 1.9e-3): a regression of the one-pass fp16
            # backward or the weight-gradient GEMM no longer hides inside a 5e-2 bound.  What this check can NOT resolve:
            # the oracle itself sits 1.8e-3 from the reference on `rendering_network.lin0.bias` of the full-width fixture
            # (forward_conf512_train: CPU fp32 against CPU fp32, the summation order of a tiny gradient) - on such biases
            # a difference below ~2e-3 is within the oracle's own distance from the reference.  The bounds of the
            # replicated embedding ('-dense': 6e-3) follow what it measures - 4.9e-3 on the radiance net's first layers - and
            # round 5's per-parameter table (profiles/r05/grad_probe_cfg3.txt, tools/experiments/grad_probe.py) says what that
            # is: the SAME figure with the exact-fp32 MLP kernels and with either training state, i.e. not the fp16 backward,
            # and the fp32 oracle itself 2.4e-3 from the same oracle in fp64 on those parameters (1.1e-3 on the zero-padded
            # embedding, where everything measures < 1e-3): gradients that are sums with heavy cancellation.  On the trained
            # network (the configs' own stand-in now) the worst parameter measures 1.5e-3.
            # ('-tier': hit points move within ~sdf_threshold / cos and the radiance net's first-layer gradients - sums with heavy
            # cancellation, on which the fp32 oracle itself is 2.4e-3 from an fp64 one - follow: 8.2e-3 measured)
            assert rel_l2(p.grad, gref) < {'cfg2': 1.5e-3, 'cfg3': 3e-3, 'cfg3-bowl': 3e-3, 'cfg3-dense': 6e-3, 'cfg4': 6e-3,
                                           'cfg4-dense': 6e-3, 'cfg3-tier': 1.5e-2, 'cfg4-tier': 1.2e-2, 'cfg3-dense-tier': 1.5e-2,
                                           'cfg3-frame': 6e-3, 'cfg2-coarse': 1.5e-3, 'cfg2-coarse-tier': 3e-3,
                                           # (tier + fp8 corrections: 26-30 of the 3072 rays draw a neighbouring lobe of the SG mixture - another
                                           # sample of the integrand - and the light's own gradient, a sum over all of them, follows: 2.0e-2 measured)
                                           'cfg3-tier-fp8': 3e-2}[bound_key], (name, rel_l2(p.grad, gref))
    print('[%s] worst parameter-gradient rel-L2 %.2e' % (wl, worst))
    # ---- algorithmic SDF evaluations: tracer counters (primary + secondary traces) = the oracle's evaluation counts
    cnt = m.ray_tracer.counter_sum.cpu().long()
    gpu_evals = ops.algorithmic_evals(cnt, 100).sum().item()
    c = Ro.counters
    cpu_evals = sum(c.get(k, 0) for k in ('sphere_trace', 'sampler', 'bisect', 'min_sdf'))
    assert abs(gpu_evals - cpu_evals) <= 0.01 * cpu_evals, (gpu_evals, cpu_evals)
    # ---- the config's own multi-ray layout on the HIP path = the per-pixel reduction of the per-ray result
    if R > 1:
        m.ray_tracer._calls = 0
        ctx = m.trace_head(to_dev(inp))
        assert torch.equal(ctx['network_object_mask'], m.last_ray_hit)      # (same rays, same tier: the same trace)
        m.uniforms_override = uniforms.to(DEV)[ctx['network_object_mask']]
        with torch.no_grad():
            multi = m.shade_tail(ctx, torch.nonzero(ctx['network_object_mask']).flatten())
        S = inp['uv'].shape[1]
        for k in FLOAT_KEYS:
            want = out[k].detach().reshape(S, R, -1)
            want = want[:, 0] if k == 'normal_values' else want.mean(1)
            assert rel_l2(multi[k], want) < 1e-5, k
        assert torch.equal(multi['network_object_mask'], out['network_object_mask'].reshape(S, R).all(1))
    # ---- '-tier' / '-fp8': where does the looser gradient bound come from?  ("justified or tightened")
    # The same forward + loss + backward on both sides with the rays that drew ANOTHER Monte-Carlo sample (a neighbouring lobe of
    # the SG mixture, a secondary ray hitting on one side only: mc_flagged_rays) taken out of the colour terms on BOTH sides: what
    # is left is the arithmetic's own effect on the gradients - moved hit points - and it has to meet the UNTIERED bound.  The
    # rest of the looser bound is those 20-30 rays being other samples of the integrand, not an error of it.
    if tier and mc_shading:
        flagged, n_dir, n_vis = mc_flagged_rays(out, ref, m.last_ray_hit, ref['_ray_hit'])
        keep = ~flagged
        for p_ in m.parameters():
            p_.grad = None
        for v in sdo.values():
            v.grad = None
        m.ray_tracer._calls = 0
        out2 = gpu_forward_with_per_ray_draws(m, to_dev(flat), uniforms)
        ref2 = Ro.forward(flat, steps1, uniforms, steps2)
        out2['object_mask'] = out2['object_mask'] & keep.to(DEV)
        ref2['object_mask'] = ref2['object_mask'] & keep
        IDRLoss(**lc)(out2, {'rgb': gt_flat.to(DEV)})['loss'].backward()
        orr.idr_loss(ref2, gt_flat, lc)['loss'].backward()
        worst_same, worst_name = 0.0, None
        for name, p_ in m.named_parameters():
            gref = sdo[name].grad
            if gref is not None and gref.norm() > 0:
                r_ = rel_l2(p_.grad, gref)
                if r_ > worst_same:
                    worst_same, worst_name = r_, name
        untiered = {'cfg3': 3e-3, 'cfg4': 6e-3}[wl] if 'dense' not in bound_key else 6e-3
        print('[%s] gradients with the %d rays that drew another sample (%d lobe / %d secondary flag) out of the colour terms on both '
              'sides: worst %.2e (%s) - all rays: %.2e; untiered bound %.1e' % (bound_key, int(flagged.sum()), n_dir, n_vis, worst_same,
                                                                              worst_name, worst, untiered))
        assert worst_same < untiered, (bound_key, worst_name, worst_same)


@pytest.mark.parametrize('wl', ['cfg3', 'cfg4'])
def test_secondary_trace_without_the_miss_search_changes_nothing_that_is_read(wl):
    """The default secondary trace skips what only fills the outputs of rays that MISS (min-SDF search, argmin fallback):
    against model.secondary_miss_search = True (the reference's recurrences) every output is bit-identical except
    secondary_points of rays outside secondary_mask - which idr_train.py:819 masks - and it executes fewer SDF
    evaluations."""
    from nefii_amd import ops
    w = syn.WORKLOADS[wl]
    mc, sd = syn.workload_state_dict(wl, seed=0)
    inp, gt = syn.make_inputs(64, w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=2)
    flat, gt_flat, R = per_ray_layout(inp, gt)
    n_ray = flat['uv'].shape[1]
    g = torch.Generator().manual_seed(7)
    steps1, steps2 = torch.rand(100, generator=g), torch.rand(100, generator=g)
    uniforms = torch.rand(n_ray, 7, generator=g)
    outs, evals = [], []
    for keep in (False, True):
        m = build_model(mc, sd, True)
        assert m.secondary_miss_search is False
        m.secondary_miss_search = keep
        m.ray_tracer.minsdf_steps_override = [steps1, steps2]
        m.ray_tracer.collect_counters = True
        m.ray_tracer.counter_sum = None
        outs.append(gpu_forward_with_per_ray_draws(m, to_dev(flat), uniforms))
        evals.append(ops.algorithmic_evals(m.ray_tracer.counter_sum.cpu().long(), 100).sum().item())
    fast, full = outs
    mask = full['secondary_mask']
    assert mask.float().mean().item() > 0.2 and (~mask).float().mean().item() > 0.2
    for k, v in full.items():
        if v is None or k == 'secondary_points':
            continue
        assert torch.equal(fast[k], v), k
    sel = mask.expand_as(full['secondary_points'])
    assert torch.equal(fast['secondary_points'][sel], full['secondary_points'][sel])
    assert torch.isfinite(fast['secondary_points']).all()
    assert evals[0] < 0.9 * evals[1], evals
    print('[%s] algorithmic SDF evaluations %d -> %d' % (wl, evals[1], evals[0]))


@pytest.mark.parametrize('wl', ['cfg3', 'cfg4', 'cfg2'])
def test_staged_searches_change_no_output_at_full_size(wl):
    """The staged min-SDF and bracket searches (nefii_tracer_params.minsdf_lipschitz, ABI 13) on the config's FULL batch - what
    bench.py times, where the tier, the quarter-row windows and the secondary trace of the Monte-Carlo configs are all in play:
    a forward of the model with RayTracing.minsdf_staged off and on gives BIT-IDENTICAL output dicts (same uniform draws), the
    audit of the slope bound stays silent, the reference's algorithmic evaluation count is the same and fewer single-pass
    evaluations are executed."""
    from nefii_amd import ops
    w = syn.WORKLOADS[wl]
    mc, sd = syn.workload_state_dict(wl, seed=0)
    inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=3)
    g = torch.Generator().manual_seed(11)
    steps1, steps2 = torch.rand(100, generator=g), torch.rand(100, generator=g)
    outs, cnts = [], []
    for staged in (False, True):
        m = build_model(mc, sd, True)
        m.ray_tracer.trace_tier = True            # as bench.py runs the config (a per-run switch since round 6)
        m.ray_tracer.bracket_staged_eval = True   # ... and the opt-in staging of the secondary traces' bracket search with it
        m.ray_tracer.minsdf_staged = staged
        m.ray_tracer.minsdf_steps_override = [steps1, steps2]
        m.ray_tracer.collect_counters = True
        m.ray_tracer.counter_sum = None
        torch.manual_seed(5)                      # the Monte-Carlo sampler's draws
        with torch.no_grad():
            outs.append(m(to_dev(inp)))
        torch.cuda.synchronize()
        cnts.append(m.ray_tracer.counter_sum.cpu().long())
        assert not m.implicit_network.coarse_audit_events
    plain, staged = outs
    for k, v in plain.items():
        if torch.is_tensor(v) and k != 'secondary_points':
            assert torch.equal(staged[k], v), (wl, k)
    if plain.get('secondary_points') is not None:
        # (of the secondary rays that MISS nothing is read - nefii_tracer_params.unread_misses - and their points are unspecified)
        sel = plain['secondary_mask'].expand_as(plain['secondary_points'])
        assert torch.equal(staged['secondary_points'][sel], plain['secondary_points'][sel]), wl
        assert torch.isfinite(staged['secondary_points']).all()
    c0, c1 = cnts
    assert c0[:, 11].sum() == 0 and c1[:, 11].sum() > 0 and c1[:, 12].max() == 0
    assert ops.algorithmic_evals(c1, 100).sum() == ops.algorithmic_evals(c0, 100).sum()
    (s0, e0), (s1, e1) = ops.executed_evals(c0, 100), ops.executed_evals(c1, 100)
    print('[%s full size] single-pass evaluations %d -> %d (x %.2f), split-precision %d -> %d' % (
        wl, e0.sum().item(), e1.sum().item(), e1.sum().item() / e0.sum().item(), s0.sum().item(), s1.sum().item()))
    assert e1.sum() < 0.8 * e0.sum() and abs(s1.sum().item() - s0.sum().item()) <= 0.02 * s0.sum().item()


@pytest.mark.parametrize('wl', ['cfg2', 'cfg3', 'cfg4'])
def test_config_full_size_properties(wl):
    """One training step (TrainStep: forward, IDRLoss, backward, 2 x Adam, secondary-consistency step where the conf has
    one) of the config at its FULL size - what bench.py times - checked through size-independent properties."""
    from nefii_amd.training.step import TrainStep
    w = syn.WORKLOADS[wl]
    mc, sd = syn.workload_state_dict(wl, seed=0)
    lc = syn.loss_conf(w['model'])
    indirect = mc.get('render_type', 'sg') != 'sg'
    inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1)
    m = build_model(mc, sd, True)
    m.ray_tracer.trace_tier = True                # as bench.py runs the config (a per-run switch since round 6)
    m.ray_tracer.collect_counters = True
    m.ray_tracer.counter_sum = None
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    step = TrainStep(m, lc, secondary_train_interval=10 if indirect else 0, secondary_batch_size=1024,
                     num_rays=w['num_rays'])
    out, lo = step(to_dev(inp), {'rgb': gt.to(DEV)})
    torch.cuda.synchronize()
    n_px = w['num_pixels']
    n_ray = n_px * max(w['num_rays'], 1)
    for k in FLOAT_KEYS:
        assert out[k].shape[0] == n_px and torch.isfinite(out[k]).all(), k
    assert all(math.isfinite(v.item()) for v in lo.values())
    hit = m.last_ray_hit.float().mean().item()
    assert m.last_ray_hit.numel() == n_ray
    assert (0.10 < hit < 0.30) if w.get('scene') is None else (0.35 < hit < 0.55), hit
    # hit points lie on the surface; hit pixels carry unit normals and colours in range
    pm = out['network_object_mask']
    assert out['sdf_output'][pm].abs().max().item() < 1e-3
    assert (out['normal_values'][pm].norm(dim=-1) - 1).abs().max().item() < 1e-3 or w['num_rays'] > 0
    assert out['sg_diffuse_albedo_values'].min().item() >= 0 and out['sg_diffuse_albedo_values'].max().item() <= 1.0001
    assert out['sg_rgb_values'].min().item() >= 0
    cnt = m.ray_tracer.counter_sum.cpu().long()
    evals = ops.algorithmic_evals(cnt, 100).sum().item()
    assert 50 * n_ray < evals < 400 * n_ray, evals / n_ray          # ~100 (closed form) .. ~250 (with secondary rays)
    if indirect:
        sec = out['secondary_mask'].float().mean().item()
        assert sec > 0.2, sec                                        # the non-convex stand-in: secondary rays re-hit
        assert out['secondary_points'].shape == (3, int(m.last_ray_hit.sum()), 3)
    after = m.state_dict()
    moved = [k for k in before if before[k].dtype.is_floating_point and not torch.equal(before[k], after[k])]
    assert any(k.startswith('envmap_material_network') for k in moved)
    assert not any(k.startswith('implicit_network') for k in moved)
    assert all(torch.isfinite(v).all() for v in after.values() if v.dtype.is_floating_point)
    print('[%s full size] %d rays, hit fraction %.3f, %.1f SDF evaluations per primary ray%s' % (
        wl, n_ray, hit, evals / n_ray, ', secondary hit fraction %.3f' % sec if indirect else ''))


def test_config5_render_strip_vs_oracle():
    """Config 5 (render.py: eval mode, conf.conf model, 256 rays per pixel, frame in raster order): 16 pixels of row 400
    of the 800 x 800 frame, straddling the object's silhouette (the bowl's rim leaves the row at column 712).  Per ray against the oracle (eval mode: no min-SDF search), then through render_frame in
    chunks (memory_capacity_level 10 = 4 pixels per chunk at 256 rays) against the un-chunked forward."""
    from nefii_amd.training import render as RR
    w = syn.WORKLOADS['cfg5']
    mc, sd = syn.workload_state_dict('cfg5', seed=0)
    full = syn.frame_inputs(w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], rows=(400, 1))
    cols = slice(704, 720)
    inp = {'uv': full['uv'][:, cols].contiguous(), 'object_mask': full['object_mask'][:, cols].contiguous(),
           'pose': full['pose'], 'intrinsics': full['intrinsics']}
    assert inp['uv'].shape == (1, 16, 256, 2) and inp['uv'][0, 0, :, 1].round().eq(400).all()
    flat, _, R = per_ray_layout(inp)
    n_ray = flat['uv'].shape[1]
    g = torch.Generator().manual_seed(7)
    uniforms = torch.rand(n_ray, 7, generator=g)
    Ro = orr.Renderer({k: v.clone() for k, v in sd.items()}, mc, training=False)
    Ro.dead_work = False
    with torch.no_grad():
        ref = Ro.forward(flat, None, uniforms, None)
    m = build_model(mc, sd, False)
    with torch.no_grad():
        out = gpu_forward_with_per_ray_draws(m, to_dev(flat), uniforms)
    compare_outputs(out, ref, max_flips=4, what='cfg5 strip', rays_per_pixel=1, ray_hit=m.last_ray_hit,
                    ref_ray_hit=ref['_ray_hit'], max_explained_frac=0.02)
    assert 0.2 < ref['_ray_hit'].float().mean().item() < 0.95 and ref['secondary_mask'].float().mean().item() > 0.0
    # chunked frame path on the same pixels (fresh sampler draws per chunk: compare what does not pass through them)
    merged = RR.render_frame(m, to_dev(inp), 16, num_rays=256, memory_capacity_level=10)
    assert merged['points'].shape == (16, 3)
    per_px = lambda k: out[k].reshape(16, R, -1)
    assert torch.equal(merged['network_object_mask'], out['network_object_mask'].reshape(16, R).all(1))
    for k in ('points', 'sg_diffuse_albedo_values', 'sg_roughness_values', 'idr_rgb_values'):
        assert rel_l2(merged[k], per_px(k).mean(1)) < 1e-5, k
    assert rel_l2(merged['normal_values'], per_px('normal_values')[:, 0]) < 1e-5
    a, b = merged['sg_rgb_values'].mean().item(), out['sg_rgb_values'].mean().item()
    assert abs(a - b) < 0.05 * abs(b), (a, b)           # other draws, same estimator: 4096 rays x 3 samples


@pytest.mark.parametrize('tier', [False, True])
def test_config5_scattered_pixels_of_the_frame_vs_oracle(tier):
    """tier: RayTracing.trace_tier, the arithmetic bench.py renders the frame with (same bounds on RGB / albedo / hit points; the
    auxiliary channels at the tier's 4e-3).  Config 5's frame at its real geometry (800 x 800, 256 rays per pixel, the frame's own sub-pixel jitter): 16 pixels
    scattered over the WHOLE frame - half of them inside the object's projection, half anywhere (silhouette, background) -
    per ray against the oracle with injected sampler draws, zero hit-mask flips tolerated beyond the knife-edge allowance,
    then through render_frame at the frame's own memory_capacity_level against the per-ray forward on everything that does
    not pass through the sampler.  (tools/render_full_frame.py does the same on 64 pixels of a whole rendered frame; this
    is the part of it that fits a test: ~20 s of oracle.)"""
    from nefii_amd.training import render as RR
    w = syn.WORKLOADS['cfg5']
    mc, sd = syn.workload_state_dict('cfg5', seed=0)
    H, W = w['image_hw']
    full = syn.frame_inputs(w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'])
    g = torch.Generator().manual_seed(5)
    # the bowl (radius 0.8 at distance 2.4, focal 1111 px) projects to ~370 px around the centre: a disc of 250 px is on it
    rad, ang = 250.0 * torch.rand(8, generator=g).sqrt(), 6.2831853 * torch.rand(8, generator=g)
    on = ((H // 2 + rad * ang.sin()).long() * W + (W // 2 + rad * ang.cos()).long())
    pick = torch.cat([on, torch.randperm(H * W, generator=g)[:8]]).unique()
    P = pick.numel()
    sub = {'uv': full['uv'][:, pick].contiguous(), 'object_mask': full['object_mask'][:, pick].contiguous(),
           'pose': full['pose'], 'intrinsics': full['intrinsics']}
    flat, _, R = per_ray_layout(sub)
    n_ray = flat['uv'].shape[1]
    assert R == 256 and n_ray == P * 256
    uniforms = torch.rand(n_ray, 7, generator=g)
    Ro = orr.Renderer({k: v.clone() for k, v in sd.items()}, mc, training=False)
    Ro.dead_work = False
    with torch.no_grad():
        ref = Ro.forward(flat, None, uniforms, None)
    m = build_model(mc, sd, False)
    m.ray_tracer.trace_tier = bool(tier)
    with torch.no_grad():
        out = gpu_forward_with_per_ray_draws(m, to_dev(flat), uniforms)
    stats = compare_outputs(out, ref, max_flips=2, what='cfg5 scattered pixels%s' % (', tier' if tier else ''), rays_per_pixel=1,
                            ray_hit=m.last_ray_hit, ref_ray_hit=ref['_ray_hit'], max_explained_frac=0.02,
                            tol_aux=4e-3 if tier else None, miss_sdf_max=None if tier else 5e-3)
    hit = ref['_ray_hit'].float().mean().item()
    print('[cfg5 scattered] %d pixels, %d rays, hit fraction %.3f, %s' % (P, n_ray, hit, stats))
    assert 0.3 < hit < 0.95 and ref['secondary_mask'].float().mean().item() > 0.0
    both = (out['network_object_mask'].cpu() == ref['network_object_mask'])
    assert rel_l2(out['sg_rgb_values'][both.to(DEV)], ref['sg_rgb_values'][both]) < 1e-3
    assert rel_l2(out['sg_diffuse_albedo_values'][both.to(DEV)], ref['sg_diffuse_albedo_values'][both]) < 1e-3
    # the frame path (chunks of 2^18 / 256 = 1024 pixels, fresh sampler draws) on the same pixels
    merged = RR.render_frame(m, to_dev(sub), P, num_rays=256, memory_capacity_level=w['memory_capacity_level'])
    per_px = lambda k: out[k].reshape(P, R, -1)
    assert torch.equal(merged['network_object_mask'], out['network_object_mask'].reshape(P, R).all(1))
    for k in ('points', 'sg_diffuse_albedo_values', 'sg_roughness_values', 'idr_rgb_values'):
        assert rel_l2(merged[k], per_px(k).mean(1)) < 1e-5, k
    assert rel_l2(merged['normal_values'], per_px('normal_values')[:, 0]) < 1e-5
    a, b = merged['sg_rgb_values'].mean().item(), out['sg_rgb_values'].mean().item()
    assert abs(a - b) < 0.05 * abs(b), (a, b)
