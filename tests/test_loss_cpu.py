"""IDRLoss (sync-free masked formulation) against the oracle's restatement of the reference loss, on CPU."""
import pytest
import torch

from nefii_amd import synthetic as syn
from nefii_amd.model.loss import IDRLoss
from oracle.renderer import idr_loss


@pytest.mark.parametrize('name', ['physg', 'conf'])
@pytest.mark.parametrize('case', ['mixed', 'all_hit', 'none_hit'])
def test_loss_matches_oracle(name, case):
    g = torch.Generator().manual_seed(3)
    n = 256
    net = torch.rand(n, generator=g) < 0.5
    obj = torch.rand(n, generator=g) < 0.7
    if case == 'all_hit':
        net[:] = True
        obj[:] = True
    if case == 'none_hit':
        net[:] = False
    out = {'network_object_mask': net, 'object_mask': obj, 'grad_theta': None,
           'idr_rgb_values': torch.rand(n, 3, generator=g).requires_grad_(True),
           'sg_rgb_values': torch.rand(n, 3, generator=g).requires_grad_(True),
           'sdf_output': torch.randn(n, 1, generator=g) * 0.1,
           'normal_values': torch.randn(n, 3, generator=g).requires_grad_(True)}
    gt = torch.rand(1, n, 3, generator=g)
    lc = syn.loss_conf(name)
    a = IDRLoss(**lc)(out, {'rgb': gt})
    b = idr_loss(out, gt, lc)
    for k in ('loss', 'idr_rgb_loss', 'sg_rgb_loss', 'mask_loss', 'normalsmooth_loss', 'background_rgb_loss'):
        assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-7), (k, a[k].item(), b[k].item())
    if not b['loss'].requires_grad:      # nothing differentiable left (no hits, no background term)
        return
    ga = torch.autograd.grad(a['loss'], [out['sg_rgb_values'], out['normal_values']], allow_unused=True)
    gb = torch.autograd.grad(b['loss'], [out['sg_rgb_values'], out['normal_values']], allow_unused=True)
    for x, y in zip(ga, gb):
        if y is None:
            assert x is None or x.abs().max() == 0
        else:
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-8)


def test_train_step_iteration_hooks_follow_the_reference_loop():
    """idr_train.py:692-713,799-802: alpha milestones, roughness/specular warm-up flags, two MultiStepLR schedulers -
    host logic of TrainStep, checked on a CPU-resident model (no kernel is launched)."""
    from nefii_amd import conf, synthetic as syn
    from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
    from nefii_amd.training.step import TrainStep
    mc = syn.model_conf('conf', hidden=64)
    m = IDRNetwork(conf.from_dict(mc))
    m.freeze_geometry()
    lc = syn.loss_conf('conf')
    a0 = lc['alpha']
    st = TrainStep(m, lc, idr_lr=5e-4, sg_lr=1e-3, idr_sched_milestones=[3, 6], idr_sched_factor=0.5,
                   sg_sched_milestones=[4], sg_sched_factor=0.1, alpha_milestones=[2, 5], alpha_factor=2.0,
                   roughness_warmup=3, specular_warmup=1)
    mat = m.envmap_material_network
    seen = []
    for it in range(8):
        st._pre_iteration()
        seen.append((st.loss.alpha, mat.fake_roughness, mat.fake_specular, st.idr_optimizer.param_groups[0]['lr'],
                     st.sg_optimizer.param_groups[0]['lr']))
        st.idr_optimizer.step()
        st.sg_optimizer.step()
        st._post_iteration()
    alphas = [s[0] for s in seen]
    assert alphas == [a0, a0, 2 * a0, 2 * a0, 2 * a0, 4 * a0, 4 * a0, 4 * a0]
    assert [s[1] for s in seen] == [True, True, True, False, False, False, False, False]
    assert [s[2] for s in seen] == [True, False, False, False, False, False, False, False]
    lr = [round(float(s[3]) / 5e-4, 6) for s in seen]
    assert lr == [1, 1, 1, 0.5, 0.5, 0.5, 0.25, 0.25]
    assert [round(float(s[4]) / 1e-3, 6) for s in seen] == [1, 1, 1, 1, 0.1, 0.1, 0.1, 0.1]
    # resuming at iteration 6 re-applies the alpha milestones already passed (idr_train.py:325-327)
    st2 = TrainStep(m, lc, alpha_milestones=[2, 5], alpha_factor=2.0, start_iter=6)
    assert st2.loss.alpha == 4 * a0


def test_min_sdf_schedule_sets_the_tracers_switches_for_the_enqueue_only():
    """TrainStep(min_sdf_every=E) used to leave skip_min_sdf_search / draw_when_skipped set on the SHARED ray
    tracer - a later TrainStep, or a direct training-mode model(...) call, inherited skip = True.  The switches now exist only
    inside the context that encloses a trace's enqueue, for the iterations the schedule names, and whatever the caller had set
    comes back (bench.py's side measurement sets skip itself); E <= 1 and trainable geometry touch nothing."""
    from nefii_amd import conf, synthetic as syn
    from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
    from nefii_amd.training.step import TrainStep
    mc = syn.model_conf('physg', hidden=64)
    m = IDRNetwork(conf.from_dict(mc))
    m.freeze_geometry()
    rt = m.ray_tracer
    st = TrainStep(m, syn.loss_conf('physg'), min_sdf_every=5)
    seen = {}
    for it in (0, 1, 4, 5, 7):
        with st._min_sdf_schedule(it):
            seen[it] = (rt.skip_min_sdf_search, rt.draw_when_skipped)
        assert (rt.skip_min_sdf_search, rt.draw_when_skipped) == (False, False)
    assert seen == {0: (False, True), 1: (True, True), 4: (True, True), 5: (False, True), 7: (True, True)}
    with st._min_sdf_schedule(6, 7, 8):             # batches traced as one call: none of them reports
        assert rt.skip_min_sdf_search is True
    with st._min_sdf_schedule(9, 10, 11):           # iteration 10 reports: the search runs for the group
        assert rt.skip_min_sdf_search is False
    rt.skip_min_sdf_search = True                   # a caller's own setting survives
    with st._min_sdf_schedule(5):
        assert rt.skip_min_sdf_search is False
    assert rt.skip_min_sdf_search is True
    st1 = TrainStep(m, syn.loss_conf('physg'))      # the default: the reference's schedule, nothing is touched
    assert st1.min_sdf_every == 1
    with st1._min_sdf_schedule(3):
        assert rt.skip_min_sdf_search is True and rt.draw_when_skipped is False
    rt.skip_min_sdf_search = False
    m.unfreeze_geometry()
    with st._min_sdf_schedule(1):
        assert rt.skip_min_sdf_search is False


def test_counter_sums_keep_the_audit_column_a_maximum_and_the_parameter_list_cache_notices_replacements():
    """ADVICE r4 lows.  (1) Column 8 of the tracer's counters holds the BITS of a float (the audit's largest difference):
    summed over stream groups or traces it must take the maximum, every other column adds.  (2) ops.param_list caches
    list(module.parameters()); a replaced Parameter object (load_state_dict(assign=True), remove_weight_norm) must not leave
    the version checks watching dead objects."""
    from nefii_amd import ops
    f = lambda x: torch.tensor([x], dtype=torch.float32).view(torch.int32).item()
    from nefii_amd import _lib
    C = _lib.TRACE_COUNTERS
    a = torch.zeros(2, 3, C, dtype=torch.int32)
    a[0, 1, 0], a[1, 1, 0] = 5, 7
    a[0, 1, 8], a[1, 1, 8] = f(2.5e-4), f(6.0e-4)
    a[0, 2, 9], a[1, 2, 9] = 10, 1
    a[0, 2, 11], a[1, 2, 11] = 3, 4
    a[0, 2, 12], a[1, 2, 12] = f(2.0e-3), f(1.0e-3)      # (ABI 13: the slope bound's audit, a maximum as well)
    s = ops.sum_counters(a)
    assert s.shape == (3, C) and s[1, 0] == 12 and s[2, 9] == 11 and s[2, 11] == 7
    assert s[1, 8].to(torch.int32).view(torch.float32).item() == pytest.approx(6.0e-4)
    assert s[2, 12].to(torch.int32).view(torch.float32).item() == pytest.approx(2.0e-3)
    b = torch.zeros(3, C, dtype=torch.int64)
    b[1, 0], b[1, 8] = 1, f(9.0e-4)
    t = ops.sum_counters(s, b)
    assert t[1, 0] == 13 and t[1, 8].to(torch.int32).view(torch.float32).item() == pytest.approx(9.0e-4)
    assert ops._audit_of(t.unsqueeze(0).to(torch.int32)) == pytest.approx(9.0e-4)
    assert ops._lip_audit_of(t.unsqueeze(0).to(torch.int32)) == pytest.approx(2.0e-3)
    # executed work: the staged search's second-stage depths count as single-pass evaluations and as waiting work
    _, coarse = ops.executed_evals(t, 100)
    assert coarse[2] == 11 + 7 and 11 in ops._WORK
    lin = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))
    p0 = ops.param_list(lin)
    assert ops.param_list(lin) is p0 and len(p0) == 4
    lin[0].weight = torch.nn.Parameter(torch.zeros(4, 4))          # the FIRST parameter replaced: noticed at once
    p1 = ops.param_list(lin)
    assert p1 is not p0 and p1[0] is lin[0].weight
    lin[1].bias = torch.nn.Parameter(torch.zeros(2))               # a later one: noticed by the periodic full check
    for _ in range(300):
        p2 = ops.param_list(lin)
    assert p2[3] is lin[1].bias


def test_slope_bound_audit_bookkeeping():
    """Host side of the staged searches' audit (ImplicitNetwork.note_lipschitz_audit): nothing happens without a violation or
    without a claim; a violation warns, records the event TrainStep re-traces on and turns the claim off for these weights; and
    the tracer's callback does not blame the slope bound for a trace whose coarse bound failed (its premise |coarse - exact| <
    tau is then void)."""
    import warnings
    from nefii_amd import conf, synthetic as syn
    from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
    mc = syn.model_conf('physg', hidden=64)
    net = IDRNetwork(conf.from_dict(mc)).implicit_network
    net._pm_version = 7
    net.note_lipschitz_audit(1e-2, 2.0)                 # no claim on record: ignored
    assert not net.coarse_audit_events
    net._lip = (7, 1.0, 2.0)
    net.note_lipschitz_audit(0.0, 2.0)                  # the bound held
    net.note_lipschitz_audit(1e-2, 0.0)                 # a trace that did not use it
    assert not net.coarse_audit_events and net._lip[2] == 2.0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        net.note_lipschitz_audit(3e-3, 2.0)
    assert any('staged min-SDF search' in str(x.message) for x in w)
    assert net.coarse_audit_events == [('lipschitz_disabled', pytest.approx(3e-3), 2.0)] and net._lip == (7, 1.0, 0.0)
    net.note_lipschitz_audit(3e-3, 2.0)                 # already off: no second event
    assert len(net.coarse_audit_events) == 1
    net._lip = (6, 1.0, 2.0)                            # a claim about OTHER weights (stale version): ignored
    net.note_lipschitz_audit(3e-3, 2.0)
    assert len(net.coarse_audit_events) == 1
