"""SampleNetwork placeholder (reference code/model/sample_network.py:10-24).

IDR's differentiable hit-point reparameterisation is only used when geometry is trainable, which no shipped
Step-2 script does (--freeze_geometry); SURVEY.md section 8a row S1 marks it inactive.  Kept as a tiny torch module so
the attribute exists; it is never on the hot path."""
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
