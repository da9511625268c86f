"""Full-frame rendering with the reference's chunk / shard / gather contract (code/scripts/render.py:267-360,
code/utils/general.py:24-37,68-82,100-107).

  * the frame is cut into chunks of ``2^level // num_rays`` pixels in raster order (split_input);
  * with W ranks: level -= floor(log2 W); the chunk list is re-ordered round-robin ``[c[i::W] for i in range(W)]``
    and cut contiguously per rank (scatter_list), exactly as render.py:284-295;
  * every rank runs the eval forward per chunk; rank 0 receives the results.  The reference pickles a list of dicts
    of tensors through gather_object; here each chunk is packed into ONE fp32 matrix [pixels, 27] and ranks exchange
    a single fixed-shape tensor with dist.gather (RCCL over xGMI on the GPU box, gloo in the CPU tests);
  * rank 0 undoes the permutation and merges (merge_output)."""
import math

import torch
import torch.distributed as dist

from ..utils import general as utils

RENDER_KEYS = [('points', 3), ('idr_rgb_values', 3), ('sg_rgb_values', 3), ('network_object_mask', 1),
               ('object_mask', 1), ('normal_values', 3), ('sg_diffuse_albedo_values', 3),
               ('sg_diffuse_rgb_values', 3), ('sg_specular_rgb_values', 3), ('sg_roughness_values', 1),
               ('sg_specular_reflection_values', 3)]
PACK_WIDTH = sum(w for _, w in RENDER_KEYS)


def plan_chunks(n_chunks, world_size):
    """-> (order, slices): order[j] = original chunk index at remapped position j (render.py:289-292),
    slices[r] = (lo, hi) of the remapped list owned by rank r (general.py:100-107)."""
    order = []
    for i in range(world_size):
        order += list(range(i, n_chunks, world_size))
    sub = n_chunks // world_size
    slices = [(r * sub, r * sub + sub if r < world_size - 1 else n_chunks) for r in range(world_size)]
    return order, slices


def pack_chunk(out):
    cols = []
    for k, w in RENDER_KEYS:
        v = out[k].detach()
        cols.append(v.reshape(v.shape[0], -1).to(torch.float32))
    return torch.cat(cols, dim=1)


def unpack_chunk(mat):
    res, c = {}, 0
    for k, w in RENDER_KEYS:
        v = mat[:, c:c + w]
        c += w
        if k in ('network_object_mask', 'object_mask'):
            v = v[:, 0] > 0.5
        res[k] = v
    return res


def render_frame(model, model_input, total_pixels, num_rays=1, memory_capacity_level=18, rank=0, world_size=1,
                 group=None):
    """Eval-mode forward of a whole frame; returns the merged output dict on rank 0 (None elsewhere)."""
    level = memory_capacity_level
    if world_size > 1:
        level -= int(math.floor(math.log2(world_size)))
    split = utils.split_input(model_input, total_pixels, num_rays, level)
    n_chunks = len(split)
    order, slices = plan_chunks(n_chunks, world_size)
    lo, hi = slices[rank]
    mine = [split[order[j]] for j in range(lo, hi)]
    packed = []
    with torch.no_grad():
        for s in mine:
            packed.append(pack_chunk(model(s)))
    batch_size = model_input['uv'].shape[0]
    if world_size == 1:
        chunks = [None] * n_chunks
        for j, m in zip(range(lo, hi), packed):
            chunks[order[j]] = m
    else:
        dev = model_input['uv'].device
        sizes = [[split[order[j]]['uv'].shape[1] * batch_size for j in range(a, b)] for a, b in slices]
        rows = max(sum(s) for s in sizes)
        buf = torch.zeros(rows, PACK_WIDTH, device=dev)
        if packed:
            cat = torch.cat(packed, dim=0)
            buf[:cat.shape[0]] = cat
        gathered = [torch.zeros_like(buf) for _ in range(world_size)] if rank == 0 else None
        dist.gather(buf, gathered, dst=0, group=group)
        if rank != 0:
            return None
        chunks = [None] * n_chunks
        for r, (a, b) in enumerate(slices):
            off = 0
            for j, sz in zip(range(a, b), sizes[r]):
                chunks[order[j]] = gathered[r][off:off + sz]
                off += sz
    res = [unpack_chunk(c) for c in chunks]
    return utils.merge_output(res, total_pixels, batch_size)
