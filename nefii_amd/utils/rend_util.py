"""Camera / ray helpers of the path (reference code/utils/rend_util.py:90-142, 200-221), on HIP.

Only the three functions the hot path uses exist here; image I/O, pose-optimisation helpers etc. are out of
scope."""
import torch

from .. import ops


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
