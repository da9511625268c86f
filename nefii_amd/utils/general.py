"""Host glue with the reference's semantics and signatures (code/utils/general.py:10-37,68-107)."""
import os
from glob import glob

import torch


def get_class(kls):
    """Dotted path -> class (the drop-in hook: conf key train.model_class, general.py:10-16)."""
    parts = kls.split('.')
    # the reference's confs name classes relative to its code/ directory (`datasets.scene_dataset.SceneDataset`,
    # `model.implicit_differentiable_renderer.IDRNetwork`): those resolve to this package's counterparts
    if parts[0] in ('datasets', 'model', 'training', 'utils'):
        parts = ['nefii_amd'] + parts
    m = __import__('.'.join(parts[:-1]))
    for comp in parts[1:]:
        m = getattr(m, comp)
    return m


def glob_imgs(path):                                             # general.py:18-22
    imgs = []
    for ext in ['*.png', '*.jpg', '*.JPEG', '*.JPG', '*.exr']:
        imgs.extend(glob(os.path.join(path, ext)))
    return imgs


def split_input(model_input, total_pixels, num_rays=1, memory_capacity_level=18):
    """Chunk a frame into pieces of 2^level // num_rays pixels, raster order (general.py:24-37)."""
    max_num = 2 ** memory_capacity_level
    n_pixels = max_num // num_rays if num_rays > 0 else max_num
    dev = model_input['uv'].device
    split = []
    for indx in torch.split(torch.arange(total_pixels, device=dev), int(n_pixels), dim=0):
        data = dict(model_input)
        data['uv'] = torch.index_select(model_input['uv'], 1, indx)
        data['object_mask'] = torch.index_select(model_input['object_mask'], 1, indx)
        split.append(data)
    return split


def merge_output(res, total_pixels, batch_size):
    """Concatenate per-chunk outputs back to [batch*total_pixels(, C)] (general.py:68-82)."""
    out = {}
    for entry in res[0]:
        if res[0][entry] is None:
            continue
        vals = [r[entry] for r in res]
        if vals[0].dim() == 1:
            out[entry] = torch.cat([v.reshape(batch_size, -1, 1) for v in vals], 1).reshape(batch_size * total_pixels)
        else:
            out[entry] = torch.cat([v.reshape(batch_size, -1, v.shape[-1]) for v in vals], 1).reshape(
                batch_size * total_pixels, -1)
    return out


def scatter_list(data_list, all_len, rank, world_size):
    """Contiguous slice for this rank; the last rank takes the remainder (general.py:100-107)."""
    sub = all_len // world_size
    if rank < world_size - 1:
        return data_list[rank * sub: rank * sub + sub]
    return data_list[rank * sub:]
