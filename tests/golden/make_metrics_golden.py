#!/usr/bin/env python3
"""tests/golden/metrics.npz: PSNR / MSE / SSIM / median-scale alignment computed by the REFERENCE's own functions
(code/scripts/evaluate.py:36-111,156-163) on seeded images (build container only).

    python tests/golden/make_metrics_golden.py

evaluate.py imports lpips and pytorch_msssim (absent) and builds an LPIPS net at import time: both are stubbed, only the
numpy functions of the file are called."""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()


class _Net:
    def __init__(self, *a, **k):
        pass

    def cuda(self):
        return self


sys.modules['lpips'] = types.ModuleType('lpips')
sys.modules['lpips'].LPIPS = _Net
sys.modules['pytorch_msssim'] = types.ModuleType('pytorch_msssim')
sys.modules['pytorch_msssim'].ssim = sys.modules['pytorch_msssim'].ms_ssim = None
spec = importlib.util.spec_from_file_location('ref_evaluate', os.path.join(ref_shim.REF_ROOT, 'scripts', 'evaluate.py'))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def main():
    g = np.random.Generator(np.random.Philox(21))
    H, W = 48, 64
    yy, xx = np.mgrid[0:H, 0:W]
    base = 0.5 + 0.4 * np.sin(xx / 5.0)[..., None] * np.cos(yy / 7.0)[..., None] * np.array([1.0, 0.8, 0.6])
    a = np.clip(base + g.normal(0, 0.05, size=(H, W, 3)), 0, 1).astype(np.float32)
    b = np.clip(base * 0.9 + g.normal(0, 0.08, size=(H, W, 3)), 0, 1).astype(np.float32)
    mask = ((yy - 24) ** 2 + (xx - 30) ** 2 < 18 ** 2)[:, :, None]
    out = {'a': a, 'b': b, 'mask': mask}
    out['psnr'] = np.float64(ref.calculate_psnr(a, b, mask))
    out['psnr_same'] = np.float64(ref.calculate_psnr(a, a, mask))
    out['mse'] = np.float64(ref.calculate_mse(a, b, mask))
    out['ssim_rgb_255'] = np.float64(ref.calculate_ssim_rgb(a.astype(np.float64) * 255, b.astype(np.float64) * 255))
    out['ssim_ch0'] = np.float64(ref.calculate_ssim(a[..., 0].astype(np.float64) * 255, b[..., 0].astype(np.float64) * 255))
    out['gauss'] = ref.matlab_style_gauss2D((11, 11), 1.5)
    gt, pre = a.copy(), (b * np.array([0.5, 2.0, 1.3], dtype=np.float32)).copy()
    pre[3:6, 3:9] = 0
    pre[20:24, 28:33, 1] = 0                    # inside the mask: the eps clamp
    ref.align_(gt, pre, mask)
    out['aligned'] = pre
    np.savez_compressed(os.path.join(HERE, 'metrics.npz'), **out)
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})


if __name__ == '__main__':
    main()
