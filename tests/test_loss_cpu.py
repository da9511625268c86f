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
