#!/usr/bin/env python3
"""Generate tests/golden/scene_instance/ (a three-view instance directory: 8-bit PNG images and masks,
cam_dict_norm.json) and tests/golden/scene_dataset.npz by RUNNING THE REFERENCE's SceneDataset on it (build container
only; code/datasets/scene_dataset.py).

    python tests/golden/make_dataset_golden.py

The reference reads images through imageio, which is not installed: ref_shim's `imageio` stub is given an `imread` that
does what imageio's Pillow plugin does for 8-bit PNGs (np.asarray of the opened image; `as_gray=True` = Pillow's mode
'F' conversion).  EXR input cannot go through the reference here (no freeimage) - the EXR reader is pinned separately
(tests/test_dataset_cpu.py).  torchvision (used by `subsample` only) is stubbed and not exercised.
"""
import json
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()


def _imread(path, as_gray=False):
    with Image.open(path) as im:
        return np.asarray(im.convert('F')) if as_gray else np.asarray(im)


sys.modules['imageio'].imread = _imread
sys.modules['torchvision'] = types.ModuleType('torchvision')
import importlib.util  # noqa: E402

# by path: `datasets` would resolve to the HuggingFace package installed in this image
_spec = importlib.util.spec_from_file_location('ref_scene_dataset', os.path.join(ref_shim.REF_ROOT, 'datasets',
                                                                                 'scene_dataset.py'))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
SceneDataset = _mod.SceneDataset  # (reference)

INST = os.path.join(HERE, 'scene_instance')
H, W, N = 12, 16, 3


def make_instance():
    g = np.random.Generator(np.random.Philox(7))
    os.makedirs(os.path.join(INST, 'image'), exist_ok=True)
    os.makedirs(os.path.join(INST, 'mask'), exist_ok=True)
    cams = {}
    for i in range(N):
        rgb = g.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        Image.fromarray(rgb).save(os.path.join(INST, 'image', 'rgb_%06d.png' % i))
        yy, xx = np.mgrid[0:H, 0:W]
        m = (((yy - H / 2 + i) ** 2 + (xx - W / 2) ** 2) < 20 + 4 * i).astype(np.uint8)
        grey = (m * 255 * (0.45 + 0.2 * g.uniform(size=(H, W)))).astype(np.uint8)     # values either side of 127.5
        Image.fromarray(np.stack([grey, grey, m * 255], -1)).save(os.path.join(INST, 'mask', 'mask_%06d.png' % i))
        K = np.eye(4)
        K[0, 0] = K[1, 1] = 20.0 + i
        K[0, 2], K[1, 2] = W / 2.0, H / 2.0
        a = 0.4 * i
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        W2C = np.eye(4)
        W2C[:3, :3] = R
        W2C[:3, 3] = [0.1 * i, -0.2, 3.0 + 0.5 * i]
        cams['rgb_%06d.png' % i] = {'K': K.reshape(-1).tolist(), 'W2C': W2C.reshape(-1).tolist(), 'img_size': [W, H]}
    with open(os.path.join(INST, 'cam_dict_norm.json'), 'w') as f:
        json.dump(cams, f, indent=1)


def main():
    make_instance()
    out = {}
    for tag, gamma, wo_mask in (('g1', 1.0, False), ('g22_womask', 2.2, True)):
        ds = SceneDataset(gamma, INST, False, 1, wo_mask)
        out[tag + '_rgb'] = torch.stack(ds.rgb_images).numpy()
        out[tag + '_mask'] = torch.stack(ds.object_masks).numpy()
        out[tag + '_K'] = torch.stack(ds.intrinsics_all).numpy()
        out[tag + '_pose'] = torch.stack(ds.pose_all).numpy()
        out[tag + '_res'] = np.array(ds.img_res + [ds.total_pixels, ds.n_cameras])
    ds = SceneDataset(1.0, INST, False, 1, False)
    idx, sample, gt = ds[1]
    out['full_uv'], out['full_rgb'] = sample['uv'].numpy(), gt['rgb'].numpy()
    np.random.seed(1)
    torch.manual_seed(2)
    ds.change_sampling_idx_patch(6, 1)
    out['patch_idx'] = ds.sampling_idx.numpy()
    ds.change_sampling_rays(4)
    out['rays'] = ds.sampling_rays.numpy()
    idx, sample, gt = ds[2]
    out['item_uv'], out['item_mask'], out['item_rgb'] = sample['uv'].numpy(), sample['object_mask'].numpy(), gt['rgb'].numpy()
    out['item_pose'], out['item_K'] = sample['pose'].numpy(), sample['intrinsics'].numpy()
    ids, s, g = ds.collate_fn([ds[0], ds[2]])
    out['coll_ids'], out['coll_uv'], out['coll_rgb'] = ids.numpy(), s['uv'].numpy(), g['rgb'].numpy()
    out['batch_rays'] = ds.batch_ray_sample(torch.arange(12.).reshape(2, 3, 2)).numpy()
    full = ds.sampling_idx.clone()
    for rank in range(3):
        ds.sampling_idx = full.clone()
        ds.scatter_sampling_idx_patch(rank, 3, 6, 1)
        out['scatter_patch_r%d' % rank] = ds.sampling_idx.numpy()
        ds.sampling_idx = full.clone()[:22]
        ds.scatter_sampling_idx(rank, 3)
        out['scatter_r%d' % rank] = ds.sampling_idx.numpy()
    torch.manual_seed(3)
    ds.change_sampling_idx(10)
    out['pixel_idx'] = ds.sampling_idx.numpy()
    np.random.seed(4)
    ds.change_sampling_idx_patch(2, 2)
    out['patch_idx_r2'] = ds.sampling_idx.numpy()
    ds.change_sampling_idx_patch(-1)
    ds.change_sampling_rays(-1)
    ds.return_single_img('rgb_000001.png')
    idx, sample, gt = ds[0]
    out['single_rgb'], out['single_uv_shape'] = gt['rgb'].numpy(), np.array(sample['uv'].shape)
    np.savez_compressed(os.path.join(HERE, 'scene_dataset.npz'), **out)
    print('wrote', len(out), 'arrays')


if __name__ == '__main__':
    main()
