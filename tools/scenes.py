"""Analytic signed-distance scenes the benchmark geometries are regressed to (measurement infrastructure: the product never
imports this).  Exact (or min / max of exact) distances near the surfaces, torch, any device.

  bowl   a ball (r 0.27) resting in a tilted bowl - the lower part of a spherical shell (mid radius 0.62, half thickness
         0.045): non-convex, so secondary rays re-hit (tools/fit_scene_sdf.py fits the 8 x 64 stand-in to the same scene)
  frame  thin features: the 12 bars (thickness 0.05) of a cube frame of half-extent 0.42, a thin tilted plate (0.024 thick)
         and a small ball inside it - rays graze bars, pass between them and re-hit others; what stresses the tracer's
         bracket search and the coarse pass's error bound
"""
import math

import torch


def bowl(p):
    n = torch.tensor([0.0, 0.55, 0.835], dtype=p.dtype, device=p.device)
    n = n / n.norm()
    r = p.norm(dim=-1)
    shell = (r - 0.62).abs() - 0.045
    cut = (p * n).sum(-1) - 0.05
    bowl_ = torch.maximum(shell, cut)
    c = -n * (0.62 - 0.045 - 0.27)
    ball = (p - c).norm(dim=-1) - 0.27
    return torch.minimum(bowl_, ball)


def _rot(ax, ay, az, dtype, device):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=dtype, device=device)
    ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=dtype, device=device)
    rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=dtype, device=device)
    return rz @ ry @ rx


def _box(p, b):
    q = p.abs() - b
    return q.clamp_min(0).norm(dim=-1) + q.max(dim=-1).values.clamp_max(0)


def _box_frame(p, b, e):
    """exact distance to the 12 edges (bars of square section 2e... e thick) of the box of half-extent b (Quilez' sdBoxFrame)"""
    p = p.abs() - b
    q = (p + e).abs() - e

    def part(x, y, z):
        v = torch.stack([x, y, z], -1)
        return v.clamp_min(0).norm(dim=-1) + torch.maximum(x, torch.maximum(y, z)).clamp_max(0)

    return torch.minimum(torch.minimum(part(p[..., 0], q[..., 1], q[..., 2]), part(q[..., 0], p[..., 1], q[..., 2])),
                         part(q[..., 0], q[..., 1], p[..., 2]))


def frame(p):
    R = _rot(0.45, 0.6, 0.2, p.dtype, p.device)
    x = p @ R
    bars = _box_frame(x, 0.42, 0.025)      # q <= 0 for p in [-2e, 0]: bars 2e = 0.05 thick
    Rp = _rot(0.9, 0.0, 0.5, p.dtype, p.device)
    plate = _box(x @ Rp, torch.tensor([0.30, 0.012, 0.30], dtype=p.dtype, device=p.device))
    ball = (x - torch.tensor([0.0, 0.16, 0.0], dtype=p.dtype, device=p.device)).norm(dim=-1) - 0.13
    return torch.minimum(torch.minimum(bars, plate), ball)


SCENES = {'bowl': bowl, 'frame': frame}
