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


# ---- what the render script writes per frame (code/scripts/render.py:361-442) -----------------------------------------
def envmap_directions(H, W, upper_hemi=False, coordinate_type='mitsuba'):
    """unit directions of an H x W latitude-longitude map (model/sg_render.py:14-33; the two axis conventions of
    sg_envmap_convention.png / blender_envmap_convention.png)"""
    top = math.pi / 2. if upper_hemi else math.pi
    if coordinate_type == 'mitsuba':
        phi, theta = torch.meshgrid([torch.linspace(0., top, H), torch.linspace(-0.5 * math.pi, 1.5 * math.pi, W)],
                                    indexing='ij')
        return torch.stack([torch.cos(theta) * torch.sin(phi), torch.cos(phi), torch.sin(theta) * torch.sin(phi)], dim=-1)
    if coordinate_type == 'blender':
        phi, theta = torch.meshgrid([torch.linspace(0., top, H), torch.linspace(1.0 * math.pi, -1.0 * math.pi, W)],
                                    indexing='ij')
        return torch.stack([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], dim=-1)
    raise ValueError('coordinate_type is mitsuba or blender, not ' + str(coordinate_type))


def compute_envmap(lgtSGs, H, W, upper_hemi=False, coordinate_type='mitsuba'):
    """sum of the light SGs over a latitude-longitude grid, [H, W, 3] (sg_render.py:10-55, envmap_type 'sg'), on the
    background-radiance kernel (nefii_env_radiance_forward, eps = 0: the lobe axes are normalised without an epsilon
    here, sg_render.py:48)"""
    from .. import ops
    dirs = envmap_directions(H, W, upper_hemi, coordinate_type).reshape(-1, 3).to(lgtSGs.device)
    with torch.no_grad():
        return ops.EnvRadianceFn.apply(lgtSGs.detach(), dirs.contiguous(), 0.0).reshape(H, W, 3)


def get_depth(points, pose):                                        # utils/rend_util.py:223-242
    """depth of [B, N, 3] world points along the camera's z axis, pose = C2W [B, 4, 4]"""
    hom = torch.cat([points, torch.ones_like(points[..., :1])], dim=2).permute(0, 2, 1)
    return torch.bmm(torch.inverse(pose), hom)[:, 2, :][:, :, None]


def frame_buffers(model, model_outputs, gt_rgb, pose, img_res):
    """merged outputs of one frame -> {name: [H, W, 3] float32 cpu} for the files render.py writes (:361-403), plus
    'panel', the tone-mapped strip of render_%03d.png (:421-430)"""
    B = gt_rgb.shape[0]
    n = gt_rgb.shape[1]

    def img(t):                                                     # plots.lin2img, then [H, W, C] of batch entry 0
        t = t.reshape(B, n, -1)
        if t.shape[-1] == 1:
            t = t.expand(B, n, 3)
        return t[0].reshape(img_res[0], img_res[1], 3).float()
    out = {'gt': img(gt_rgb), 'rerender_rgb': img(model_outputs['sg_rgb_values']),
           'diffuse_rgb': img(model_outputs['sg_diffuse_rgb_values']),
           'specular_rgb': img(model_outputs['sg_specular_rgb_values']),
           'diffuse_albedo': img(model_outputs['sg_diffuse_albedo_values']),
           'roughness': img(model_outputs['sg_roughness_values'])}
    refl = model.envmap_material_network.specular_inv_remap(model_outputs['sg_specular_reflection_values'])
    out['specular_reflection'] = img(refl)
    mask = model_outputs['network_object_mask'].reshape(-1).bool()
    depth = torch.ones(B * n, device=mask.device)
    if mask.sum() > 0:
        valid = get_depth(model_outputs['points'].reshape(B, n, 3), pose.to(mask.device)).reshape(-1)[mask]
        depth[mask] = valid
        depth[~mask] = 0.98 * valid.min()
    out['depth'] = img(depth[:, None])
    normal = img(torch.clamp((model_outputs['normal_values'] + 1.) / 2., 0., 1.))
    tone = lambda x: torch.clamp(torch.pow(x.clamp_min(0.), 1. / 2.2), 0., 1.)
    out['panel'] = torch.cat([tone(out['gt']), tone(out['rerender_rgb']), tone(out['diffuse_rgb']),
                              tone(out['specular_rgb']), normal, tone(out['diffuse_albedo']), out['roughness'],
                              out['specular_reflection']], dim=1)
    return {k: v.detach().cpu() for k, v in out.items()}


def write_frame(model, model_outputs, gt_rgb, pose, img_res, plots_dir, index):
    """the per-frame files of render.py:407-430, named as scripts/evaluate.py expects them"""
    import os
    import numpy as np
    from PIL import Image
    from ..utils import exr
    os.makedirs(plots_dir, exist_ok=True)
    buf = frame_buffers(model, model_outputs, gt_rgb, pose, img_res)
    for name in ('gt', 'rerender_rgb', 'diffuse_rgb', 'specular_rgb', 'diffuse_albedo', 'roughness', 'specular_reflection'):
        exr.imwrite(os.path.join(plots_dir, '%s-%03d.exr' % (name, index)), buf[name].numpy())
    panel = (buf['panel'].clamp(0., 1.).numpy() * 255.).astype(np.uint8)
    Image.fromarray(panel).save(os.path.join(plots_dir, 'render_%03d.png' % index))
    return buf


def write_envmap(model, plots_dir, coordinate_type='mitsuba', H=256, W=512):       # render.py:432-442
    import os
    from ..utils import exr
    net = model.envmap_material_network
    env = compute_envmap(net.get_light(), H, W, upper_hemi=getattr(net, 'upper_hemi', False),
                         coordinate_type=coordinate_type)
    os.makedirs(plots_dir, exist_ok=True)
    exr.imwrite(os.path.join(plots_dir, 'envmap.exr'), env.cpu().numpy())
    return env
