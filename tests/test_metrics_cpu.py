"""Evaluation metrics (code/scripts/evaluate.py): against values computed by the
reference's own numpy functions (tests/golden/make_metrics_golden.py), plus the directory walk of `main`."""
import os

import numpy as np
import pytest
import torch

from nefii_amd.scripts import evaluate as ev
from nefii_amd.utils import exr

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def gold():
    return dict(np.load(os.path.join(HERE, 'golden', 'metrics.npz')))


def test_psnr_mse_ssim_alignment_match_the_reference(gold):
    a, b, mask = gold['a'], gold['b'], gold['mask']
    assert abs(ev.calculate_psnr(a, b, mask) - gold['psnr']) < 1e-12
    assert ev.calculate_psnr(a, a, mask) == float('inf')
    assert abs(ev.calculate_mse(a, b, mask) - gold['mse']) < 1e-15
    # the reference's numpy SSIM on 0..255 images = SSIM with data_range 1 on 0..1 images (C1, C2 scale with L^2)
    assert abs(ev.calculate_ssim(a, b) - gold['ssim_rgb_255']) < 1e-9
    assert abs(ev.calculate_ssim(a[..., :1], b[..., :1]) - gold['ssim_ch0']) < 1e-9
    assert abs(ev.calculate_ssim(a.astype(np.float64) * 255, b.astype(np.float64) * 255, data_range=255.) - gold['ssim_rgb_255']) < 1e-9
    w = ev.gaussian_window()
    assert np.abs(torch.outer(w, w).numpy() - gold['gauss']).max() < 1e-15
    gt, pre = a.copy(), (b * np.array([0.5, 2.0, 1.3], dtype=np.float32)).copy()
    pre[3:6, 3:9] = 0
    pre[20:24, 28:33, 1] = 0
    ev.align_(gt, pre, mask)
    assert np.array_equal(pre, gold['aligned'])


def test_ms_ssim_properties():
    g = np.random.Generator(np.random.Philox(2))
    yy, xx = np.mgrid[0:176, 0:200]
    a = (0.5 + 0.4 * np.sin(xx / 9.0) * np.cos(yy / 5.0))[..., None].repeat(3, -1).astype(np.float32)
    assert abs(ev.calculate_ms_ssim(a, a) - 1.0) < 1e-12
    vals = [ev.calculate_ms_ssim(a, np.clip(a + g.normal(0, s, a.shape), 0, 1).astype(np.float32)) for s in (0.02, 0.1, 0.3)]
    assert 1 > vals[0] > vals[1] > vals[2] > 0
    # one scale of it is SSIM's contrast-structure term; a single-scale product with weight 1 is SSIM itself
    x = torch.from_numpy(a).double().permute(2, 0, 1)[None]
    s, cs = ev._ssim_cs(x, x * 0.9, 1.0, ev.gaussian_window())
    assert (s <= cs + 1e-12).all()
    with pytest.raises(ValueError):
        ev.calculate_ms_ssim(a[:100], a[:100])


def test_evaluate_directory_walk_and_results_file(tmp_path):
    from PIL import Image
    g = np.random.Generator(np.random.Philox(4))
    gt, plots = tmp_path / 'scene' / 'test', tmp_path / 'exp' / 'plots'
    for d in ('image', 'diffuse', 'roughness', 'sp_rgb', 'mask'):
        (gt / d).mkdir(parents=True)
    plots.mkdir(parents=True)
    H, W = 40, 48
    mask = np.zeros((H, W), np.uint8)
    mask[8:32, 10:40] = 255
    for i in (0, 3):
        Image.fromarray(mask).save(gt / 'mask' / ('%06d.png' % i))
        truth = {k: g.uniform(0.05, 0.9, size=(H, W, 3)).astype(np.float32) for k in ('rgb', 'diffuse', 'rough', 'sp')}
        exr.imwrite(str(gt / 'image' / ('%06d.exr' % i)), truth['rgb'])
        exr.imwrite(str(gt / 'diffuse' / ('%06d_diffuse.00.exr' % i)), truth['diffuse'])
        exr.imwrite(str(gt / 'roughness' / ('%06d.exr' % i)), truth['rough'])
        exr.imwrite(str(gt / 'sp_rgb' / ('%06d_sprgb.00.exr' % i)), truth['sp'])
        exr.imwrite(str(plots / ('rerender_rgb-%03d.exr' % i)), truth['rgb'])                       # perfect
        exr.imwrite(str(plots / ('diffuse_albedo-%03d.exr' % i)), truth['diffuse'] * np.float32(0.5))  # off by a scale
        exr.imwrite(str(plots / ('roughness-%03d.exr' % i)), truth['rough'] + np.float32(0.1))
        exr.imwrite(str(plots / ('specular_rgb-%03d.exr' % i)), truth['sp'] * np.float32(1.1))
    res = ev.main(str(plots), str(gt))
    assert set(res) == {'rgb', 'diffuse', 'diffuse_align', 'roughness', 'sp_rgb'}
    assert res['rgb']['psnr'] == float('inf') and abs(res['rgb']['ssim'] - 1) < 1e-9
    assert res['diffuse_align']['psnr'] > 100 and res['diffuse']['psnr'] < 25        # the median scale undoes the 0.5
    inside = (mask > 0).mean()
    assert abs(res['roughness']['mse'] - 0.01 * inside) < 1e-6
    assert set(res['diffuse']) == {'psnr', 'ssim', 'ms_ssim', 'lpips', 'mse'} and np.isnan(res['rgb']['lpips'])
    txt = (tmp_path / 'exp' / 'results.txt').read_text()
    assert '>>>>>>>>>>rgb        <<<<<<<<<<' in txt and 'psnr       ssim       ms_ssim    lpips' in txt


def test_envmap_grid_matches_the_reference():
    """lat-long direction grids of both axis conventions (+ upper hemisphere) and the SG sum over them, against
    sg_render.compute_envmap run by tests/golden/make_envmap_golden.py"""
    from nefii_amd.training.render import envmap_directions
    g = dict(np.load(os.path.join(HERE, 'golden', 'envmap_ref.npz')))
    l = torch.from_numpy(g['lgtSGs'])
    ax = l[:, :3] / l[:, :3].norm(dim=-1, keepdim=True)
    for ct in ('mitsuba', 'blender'):
        for hemi in (False, True):
            dirs = envmap_directions(12, 20, hemi, ct)
            sg = (l[:, -3:].abs() * torch.exp(l[:, 3:4].abs() * ((dirs[..., None, :] * ax).sum(-1, keepdim=True) - 1.))).sum(-2)
            ref = torch.from_numpy(g['%s_%d' % (ct, int(hemi))])
            assert sg.shape == ref.shape == (12, 20, 3)
            assert (sg - ref).abs().max().item() <= 2e-6 * ref.abs().max().item(), (ct, hemi)
    with pytest.raises(ValueError):
        envmap_directions(4, 8, False, 'opengl')
