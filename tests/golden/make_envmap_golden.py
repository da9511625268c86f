#!/usr/bin/env python3
"""tests/golden/envmap_ref.npz: the reference's `model/sg_render.py:compute_envmap` on seeded light SGs, both axis
conventions, full sphere and upper hemisphere (build container only).

    python tests/golden/make_envmap_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from model.sg_render import compute_envmap  # noqa: E402  (reference)


def main():
    g = torch.Generator().manual_seed(17)
    lgt = torch.randn(24, 7, generator=g)
    lgt[:, 3] = lgt[:, 3].abs() * 20 + 2
    out = {'lgtSGs': lgt.numpy()}
    for ct in ('mitsuba', 'blender'):
        for hemi in (False, True):
            env = compute_envmap(lgtSGs=lgt, H=12, W=20, upper_hemi=hemi, log=False, coordinate_type=ct)
            out['%s_%d' % (ct, int(hemi))] = env.numpy()
    np.savez_compressed(os.path.join(HERE, 'envmap_ref.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
