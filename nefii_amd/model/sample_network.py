"""SampleNetwork (reference code/model/sample_network.py:10-24): the ray / surface intersection as a differentiable
function of the implicit geometry (IDR, equation 3):  x(theta) = c + (t - (f(x; theta) - f0) / (grad f . d)) d.

Used by the trainable-geometry branch only (model/trainable_geometry.py) - a torch slow path:
every shipped Step-2 script freezes the geometry, where the hit point is a constant."""
import torch
import torch.nn as nn


class SampleNetwork(nn.Module):
    def forward(self, surface_output, surface_sdf_values, surface_points_grad, surface_dists, surface_cam_loc,
                surface_ray_dirs):
        dirs0 = surface_ray_dirs.detach()
        dot = torch.sum(surface_points_grad * dirs0, dim=-1, keepdim=True)
        dot = torch.where(dot.abs() < 1e-8, torch.full_like(dot, 1e-8), dot)
        t_theta = surface_dists - (surface_output - surface_sdf_values) / dot
        return surface_cam_loc + t_theta * surface_ray_dirs
