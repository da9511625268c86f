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
