"""Evaluation of rendered buffers against ground truth (reference code/scripts/evaluate.py:18-330, compute_psnr.py):
masked PSNR / SSIM / MS-SSIM of the re-rendered image, the diffuse albedo (raw, and after the per-channel median scale
alignment) and the specular image; MSE of albedo and roughness; the `results.txt` summary.

    python -m nefii_amd.scripts.evaluate --pre_dir <.../plots> --gt_dir <.../test>

The reference borrows SSIM / MS-SSIM from pytorch-msssim and LPIPS from the lpips package (requirements.sh:11, versions
unpinned); neither is installable here.  SSIM and MS-SSIM are restated from their published definitions as
pytorch-msssim implements them (Wang et al. 2003/2004: 11-tap Gaussian window, sigma 1.5, K = (0.01, 0.03), 'valid'
filtering, five scales weighted 0.0448 / 0.2856 / 0.3001 / 0.2363 / 0.1333 with 2x2 average pooling between them) and
pinned against the reference's own numpy SSIM (evaluate.py:57-111).  LPIPS needs the trained AlexNet + linear-head
weights, which cannot be obtained offline: the "lpips" entry is reported as nan and says so once."""
import argparse
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from ..utils import rend_util


def load_rgb(path):                                                 # evaluate.py:18-25 -> [H, W, C]
    return rend_util.load_rgb(path).transpose(1, 2, 0)


load_mask = rend_util.load_mask                                     # :28-33


def calculate_psnr(img1, img2, mask=None):                          # :36-44 (range [0, 1]; the mask is not used)
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    return float('inf') if mse == 0 else 20 * math.log10(1.0 / math.sqrt(mse))


def calculate_mse(img1, img2, mask=None):                           # :47-54
    return np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)


def gaussian_window(size=11, sigma=1.5):
    x = torch.arange(size, dtype=torch.float64) - size // 2
    g = torch.exp(-x * x / (2 * sigma * sigma))
    return g / g.sum()


def _filter(x, win):
    """separable 'valid' Gaussian filtering of [N, C, H, W]"""
    C = x.shape[1]
    k = win.to(x)
    x = F.conv2d(x, k.view(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C)
    return F.conv2d(x, k.view(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C)


def _ssim_cs(x, y, data_range, win, K=(0.01, 0.03)):
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _filter(x, win), _filter(y, win)
    s11 = _filter(x * x, win) - mu1 * mu1
    s22 = _filter(y * y, win) - mu2 * mu2
    s12 = _filter(x * y, win) - mu1 * mu2
    cs_map = (2 * s12 + C2) / (s11 + s22 + C2)
    ssim_map = (2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)          # per image, per channel


def calculate_ssim(img1, img2, data_range=1.0):
    """SSIM of [H, W, C] images, mean over channels (evaluate.py:76-111 with L = data_range; what
    pytorch_msssim.ssim(..., size_average=False) returns for one image)"""
    x = torch.from_numpy(np.ascontiguousarray(img1)).double().permute(2, 0, 1)[None]
    y = torch.from_numpy(np.ascontiguousarray(img2)).double().permute(2, 0, 1)[None]
    return _ssim_cs(x, y, data_range, gaussian_window())[0].mean().item()


MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def calculate_ms_ssim(img1, img2, data_range=1.0):
    x = torch.from_numpy(np.ascontiguousarray(img1)).double().permute(2, 0, 1)[None]
    y = torch.from_numpy(np.ascontiguousarray(img2)).double().permute(2, 0, 1)[None]
    if min(x.shape[-2:]) <= (11 - 1) * 2 ** 4:
        raise ValueError('MS-SSIM over five scales needs images larger than 160 pixels on their smaller side')
    win = gaussian_window()
    mcs = []
    for i in range(len(MS_WEIGHTS)):
        ssim_c, cs = _ssim_cs(x, y, data_range, win)
        if i < len(MS_WEIGHTS) - 1:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in x.shape[2:]]
            x, y = F.avg_pool2d(x, 2, padding=pad), F.avg_pool2d(y, 2, padding=pad)
    vals = torch.stack(mcs + [torch.relu(ssim_c)], dim=0)                   # [levels, N, C]
    w = torch.tensor(MS_WEIGHTS, dtype=vals.dtype).view(-1, 1, 1)
    return torch.prod(vals ** w, dim=0).mean().item()


_lpips_note = [False]


def calculate_lpips(img1, img2):
    if not _lpips_note[0]:
        print('[nefii_amd] LPIPS needs pretrained AlexNet weights that are not available offline: reported as nan')
        _lpips_note[0] = True
    return float('nan')


def align_(rgb_gt, rgb_pre, mask, eps=1e-4):                        # :156-163: per-channel median scale, in place
    for c in range(rgb_gt.shape[2]):
        gt_value = rgb_gt[..., c:c + 1][mask]
        pre_value = rgb_pre[..., c:c + 1][mask]
        pre_value[pre_value <= eps] = eps
        rgb_pre[..., c] *= np.median(gt_value / pre_value)


def _white_background(img, mask):
    out = img * mask
    out[~np.broadcast_to(mask, out.shape)] = 1
    return out


def evaluate_rgb(rgb_pre_path, rgb_gt_path, mask_path, align=False, tonemap=True):     # :114-153
    rgb_pre, rgb_gt = load_rgb(rgb_pre_path), load_rgb(rgb_gt_path)
    mask = load_mask(mask_path)[:, :, None]
    if tonemap:
        rgb_pre = np.clip(np.power(rgb_pre, 1. / 2.2), 0., 1.)
        rgb_gt = np.clip(np.power(rgb_gt, 1. / 2.2), 0., 1.)
    if align:
        align_(rgb_gt, rgb_pre, mask)
    pre, gt = _white_background(rgb_pre, mask), _white_background(rgb_gt, mask)
    out = {'psnr': calculate_psnr(pre, gt, mask), 'ssim': calculate_ssim(pre.astype(np.float32), gt.astype(np.float32))}
    try:
        out['ms_ssim'] = calculate_ms_ssim(pre.astype(np.float32), gt.astype(np.float32))
    except ValueError:
        out['ms_ssim'] = float('nan')
    out['lpips'] = calculate_lpips(pre, gt)
    return out


def evaluate_raw(rgb_pre_path, rgb_gt_path, mask_path):             # :166-180
    mask = load_mask(mask_path)[:, :, None]
    return {'mse': calculate_mse(load_rgb(rgb_pre_path) * mask, load_rgb(rgb_gt_path) * mask, mask)}


def main(prediction_dir, gt_path):                                  # :191-303
    sub = {k: os.path.join(gt_path, k) for k in ('image', 'diffuse', 'roughness', 'sp_rgb', 'mask')}
    all_result = {}

    def put(result, key):
        for k, v in result.items():
            all_result.setdefault(key, {}).setdefault(k, []).append(v)
    for file_name in sorted(os.listdir(sub['image'])):
        index = int(file_name.split('.')[0])
        mask = os.path.join(sub['mask'], '%06d.png' % index)
        pre = lambda stem: os.path.join(prediction_dir, '%s-%03d.exr' % (stem, index))
        put(evaluate_rgb(pre('rerender_rgb'), os.path.join(sub['image'], file_name), mask), 'rgb')
        gt_diffuse = os.path.join(sub['diffuse'], '%06d_diffuse.00.exr' % index)
        r = evaluate_rgb(pre('diffuse_albedo'), gt_diffuse, mask, tonemap=False)
        r.update(evaluate_raw(pre('diffuse_albedo'), gt_diffuse, mask))
        put(r, 'diffuse')
        put(evaluate_rgb(pre('diffuse_albedo'), gt_diffuse, mask, align=True, tonemap=False), 'diffuse_align')
        put(evaluate_raw(pre('roughness'), os.path.join(sub['roughness'], '%06d.exr' % index), mask), 'roughness')
        put(evaluate_rgb(pre('specular_rgb'), os.path.join(sub['sp_rgb'], '%06d_sprgb.00.exr' % index), mask), 'sp_rgb')
    path = os.path.join(os.path.dirname(prediction_dir), 'results.txt')
    with open(path, 'a') as fp:
        for key, res in all_result.items():
            for k in res:
                res[k] = float(np.array(res[k]).mean())
            fp.write('\n>>>>>>>>>>{}<<<<<<<<<<\n'.format(key.ljust(11, ' ')))
            fp.write(''.join(k.ljust(11, ' ') for k in res) + '\n')
            fp.write(''.join(str('%.6f' % v).ljust(11, ' ') for v in res.values()) + '\n')
    print(all_result)
    return all_result


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--pre_dir', type=str, default='', help='path to rendering folder')
    parser.add_argument('--gt_dir', type=str, default='', help='path to ground truth')
    opt = parser.parse_args()
    main(opt.pre_dir.rstrip('/'), opt.gt_dir.rstrip('/'))
