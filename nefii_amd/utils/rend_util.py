"""Camera / ray helpers of the path (reference code/utils/rend_util.py:90-142, 200-221), on HIP.

Only the three functions the hot path uses exist here; image I/O, pose-optimisation helpers etc. are out of
scope."""
import numpy as np
import torch

from .. import ops


def _imread(path):
    """imageio.imread of the reference: float32 [H, W, C] for .exr (own reader, utils/exr.py), uint8 for the rest"""
    if path.endswith('.exr'):
        from . import exr
        return exr.imread(path)
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ('L', 'RGB', 'RGBA'):
            im = im.convert('RGBA' if 'A' in im.getbands() or 'transparency' in im.info else 'RGB')
        return np.asarray(im)


def load_rgb(path):                                              # rend_util.py:13-20
    img = _imread(path)
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, axis=2)
    img = np.float32(img[:, :, :3])
    if not path.endswith('.exr'):
        img = img / 255.
    return img.transpose(2, 0, 1)                                # [C, H, W]


def load_mask(path):                                             # rend_util.py:23-28
    """`imageio.imread(path, as_gray=True)`: Pillow's mode 'F' conversion (ITU-R 601-2 luma), then > 0.5 of 255"""
    if path.endswith('.exr'):
        img = _imread(path)
        alpha = img if img.ndim == 2 else img[:, :, :3] @ np.array([0.299, 0.587, 0.114], dtype=np.float32)
    else:
        from PIL import Image
        with Image.open(path) as im:
            alpha = np.asarray(im.convert('F'))
    return np.float32(alpha) / 255. > 0.5


def get_camera_params(uv, pose, intrinsics):
    """uv [B,S,2], pose [B,4,4] (cam-to-world), intrinsics [B,4,4] -> (ray_dirs [B,S,3], cam_loc [B,3])."""
    if pose.shape[1] == 7:
        raise NotImplementedError('quaternion poses (pose optimisation is outside the hot path)')
    dirs, _ = ops.camera_rays(uv, pose, intrinsics)
    return dirs, pose[:, :3, 3].to(torch.float32)


def get_sphere_intersection(cam_loc, ray_directions, r=1.0):
    """Near/far depths of the |x| = r sphere along unit rays, clamped at 0.01; mask of rays that hit it.

    cam_loc [B,3], ray_directions [B,S,3] -> [B,S,2], [B,S].  Elementwise plumbing used by callers outside the
    tracer; the tracer kernel computes the same quantities in-kernel (nefii_tracer.hip, round 0)."""
    b = torch.sum(ray_directions * cam_loc.unsqueeze(1), dim=-1)
    under = b ** 2 - (cam_loc.norm(2, 1, keepdim=True) ** 2 - r ** 2)
    hit = under > 0
    root = torch.sqrt(torch.clamp(under, min=0.0))
    t = torch.stack([-root - b, root - b], dim=-1)
    t = torch.where(hit.unsqueeze(-1), t, torch.zeros_like(t)).clamp_min(0.01)
    return t, hit
