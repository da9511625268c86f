#!/usr/bin/env python3
"""tests/golden/general_ref.npz: the reference's chunking helpers (code/utils/general.py:24-37,68-82,100-107) and the
round-robin re-ordering of render.py:286-295 on a small frame (build container only).

    python tests/golden/make_general_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from utils import general as ref  # noqa: E402  (reference)


def main():
    g = torch.Generator().manual_seed(5)
    out = {}
    for tag, total, num_rays, level, batch in (('single', 77, -1, 4, 1), ('multi', 50, 3, 5, 2)):
        uv = torch.rand(batch, total, *((num_rays, 2) if num_rays > 0 else (2,)), generator=g)
        mask = torch.rand(batch, total, generator=g) > 0.5
        inp = {'uv': uv, 'object_mask': mask, 'pose': torch.eye(4)[None].repeat(batch, 1, 1)}
        split = ref.split_input(inp, total, num_rays, level)
        out[tag + '_uv'], out[tag + '_mask'] = uv.numpy(), mask.numpy()
        out[tag + '_sizes'] = np.array([s['uv'].shape[1] for s in split])
        out[tag + '_first_uv'] = split[1]['uv'].numpy()
        # per-chunk "outputs": a 1-D and a 2-D entry, merged back
        res = [{'a': s['uv'].reshape(batch, s['uv'].shape[1], -1).sum(-1).reshape(-1),
                'b': s['uv'].reshape(batch, s['uv'].shape[1], -1)[..., :2].reshape(-1, 2), 'none': None} for s in split]
        merged = ref.merge_output(res, total, batch)
        out[tag + '_merged_a'], out[tag + '_merged_b'] = merged['a'].numpy(), merged['b'].numpy()
        assert 'none' not in merged
        for world in (2, 3):
            order = []
            for i in range(world):
                order += list(range(len(split)))[i:len(split):world]
            out['%s_order_w%d' % (tag, world)] = np.array(order)
            for rank in range(world):
                out['%s_scatter_w%d_r%d' % (tag, world, rank)] = np.array(ref.scatter_list(order, len(order), rank, world))
    np.savez_compressed(os.path.join(HERE, 'general_ref.npz'), **out)
    print(len(out), 'arrays')


if __name__ == '__main__':
    main()
