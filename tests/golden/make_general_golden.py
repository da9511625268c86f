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
    # BASELINE config 5 (robot/render.sh: 800 x 800 frame, num_rays 256, memory_capacity_level 18) at W = 1, 2, 8 ranks:
    # chunk sizes of split_input at level 18 - floor(log2 W) (render.py:284-286), the round-robin order and every rank's
    # slice of it (:289-295).  The uv tensor carries no ray dimension here (split_input only indexes pixels; the chunk
    # size comes from its n_rays argument) - 640 000 x 256 x 2 floats would be 1.3 GB.
    total, num_rays, level = 800 * 800, 256, 18
    uv = torch.zeros(1, total, 2)
    uv[0, :, 0] = torch.arange(total)
    inp = {'uv': uv, 'object_mask': torch.ones(1, total, dtype=torch.bool)}
    for world in (1, 2, 8):
        lv = level - int(np.floor(np.log2(world)))
        split = ref.split_input(inp, total, num_rays, lv)
        sizes = np.array([s['uv'].shape[1] for s in split])
        first = np.array([int(s['uv'][0, 0, 0]) for s in split])
        out['cfg5_w%d_sizes' % world] = sizes.astype(np.int32)
        out['cfg5_w%d_first_pixel' % world] = first.astype(np.int32)
        order = []
        for i in range(world):
            order += list(range(len(split)))[i:len(split):world]
        out['cfg5_w%d_order' % world] = np.array(order, dtype=np.int32)
        parts = [ref.scatter_list(order, len(order), rank, world) for rank in range(world)]
        out['cfg5_w%d_scatter_lens' % world] = np.array([len(p) for p in parts], dtype=np.int32)
        out['cfg5_w%d_scatter' % world] = np.concatenate([np.array(p, dtype=np.int32) for p in parts])
    np.savez_compressed(os.path.join(HERE, 'general_ref.npz'), **out)
    print(len(out), 'arrays')


if __name__ == '__main__':
    main()
