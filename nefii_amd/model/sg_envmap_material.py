"""EnvmapMaterialNetwork with the reference's constructor, parameters and forward contract
(code/model/sg_envmap_material.py:46-447); the albedo(+roughness) MLP runs as one fused HIP kernel.

State-dict keys kept: ``diffuse_albedo_layers.{0,2,...}.weight/.bias``, ``lgtSGs``, ``specular_reflectance``,
``roughness``.  Supported option sets are the shipped ones: conf.conf (same_mlp + roughness_mlp + specular_mlp +
fix_specular_albedo) and physg.conf (global roughness / specular).  Options no shipped conf enables
(num_base_materials > 1, separate roughness/specular MLPs, use_normal, 2-D envmap light, correct_normal)
raise NotImplementedError."""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def fibonacci_sphere(samples=1):
    """Evenly spread lobe axes for the initial light (sg_envmap_material.py:19-38)."""
    i = np.arange(samples, dtype=np.float64)
    y = 1 - (i / float(samples - 1)) * 2
    radius = np.sqrt(1 - y * y)
    theta = np.pi * (3. - np.sqrt(5.)) * i
    return np.stack([np.cos(theta) * radius, y, np.sin(theta) * radius], -1)


def compute_energy(lgtSGs):
    lam = torch.abs(lgtSGs[:, 3:4])
    mu = torch.abs(lgtSGs[:, 4:])
    return mu * 2.0 * np.pi / lam * (1.0 - torch.exp(-2.0 * lam))


class EnvmapMaterialNetwork(nn.Module):
    def __init__(self, multires=0, dims=[256, 256, 256], white_specular=False, white_light=False, num_lgt_sgs=32,
                 num_base_materials=2, upper_hemi=False, fix_specular_albedo=False, specular_albedo=[-1., -1., -1.],
                 init_specular_reflectance=-1, correct_normal=False, roughness_mlp=False, specular_mlp=False,
                 same_mlp=False, dims_roughness=[256, 256, 256], dims_specular=[256, 256, 256],
                 feature_vector_size=0, use_normal=False, light_type='sg'):
        super().__init__()
        if correct_normal or use_normal or light_type != 'sg' or num_base_materials != 1:
            raise NotImplementedError('EnvmapMaterialNetwork option not used by any shipped conf')
        if (roughness_mlp or (specular_mlp and not fix_specular_albedo)) and not same_mlp:
            raise NotImplementedError('separate roughness / specular MLPs (no shipped conf)')
        self.roughness_mlp = roughness_mlp
        self.specular_mlp = specular_mlp
        self.same_mlp = same_mlp
        self.feature_vector_size = feature_vector_size
        self.fix_specular_albedo = fix_specular_albedo
        self.fake_roughness = False
        self.fake_specular = False
        self.light_type = light_type
        self.multires = multires
        self.use_normal = use_normal
        self.white_light = white_light
        self.white_specular = white_specular
        self.upper_hemi = upper_hemi
        self.numLgtSGs = num_lgt_sgs
        self.numBrdfSGs = num_base_materials
        self.dim_o = 3 + (1 if (roughness_mlp and same_mlp) else 0) + \
            (1 if (not fix_specular_albedo and specular_mlp and same_mlp) else 0)
        self.cfg = dict(multires=multires, dims=list(dims))
        self.specs, self.enc = ops.material_specs(self.cfg, feature_vector_size, self.dim_o)
        layers = []
        for l, s in enumerate(self.specs):
            layers.append(nn.Linear(s.k_in, s.n_out))
            if l < len(self.specs) - 1:
                layers.append(nn.ELU())
        self.diffuse_albedo_layers = nn.Sequential(*layers)
        # light: [M,7] = lobe axis, sharpness, rgb amplitude; initialisation as :126-157
        if white_light:
            self.lgtSGs = nn.Parameter(torch.randn(num_lgt_sgs, 5), requires_grad=True)
        else:
            self.lgtSGs = nn.Parameter(torch.randn(num_lgt_sgs, 7), requires_grad=True)
            self.lgtSGs.data[:, -2:] = self.lgtSGs.data[:, -3:-2].expand((-1, 2))
        self.lgtSGs.data[:, 3:4] = 20. + torch.abs(self.lgtSGs.data[:, 3:4] * 100.)
        energy = compute_energy(self.lgtSGs.data)
        self.lgtSGs.data[:, 4:] = torch.abs(self.lgtSGs.data[:, 4:]) / torch.sum(energy, dim=0, keepdim=True) * 2. * np.pi
        self.lgtSGs.data[:, :3] = torch.from_numpy(fibonacci_sphere(num_lgt_sgs).astype(np.float32))
        if upper_hemi:
            self.lgtSGs.data = self.restrict_lobes_upper(self.lgtSGs.data)
        if fix_specular_albedo:
            sa = np.array(specular_albedo).astype(np.float32)
            assert np.all(np.logical_and(sa > 0., sa < 1.))
            self.specular_reflectance = nn.Parameter(torch.from_numpy(sa).reshape((1, 3)), requires_grad=False)
        elif not specular_mlp:
            self.specular_reflectance = nn.Parameter(torch.abs(torch.randn(1, 1 if white_specular else 3)),
                                                     requires_grad=True)
            if init_specular_reflectance > 0:
                self.specular_reflectance.data[:] = np.log(1 / (1 - init_specular_reflectance) - 1)
        if not roughness_mlp:
            r = np.array([np.random.uniform(1.5, 2.0)]).astype(np.float32).reshape((1, 1))
            self.roughness = nn.Parameter(torch.from_numpy(r), requires_grad=True)
        self._pm = None

    @staticmethod
    def restrict_lobes_upper(lgtSGs):
        return torch.cat((lgtSGs[..., :1], torch.abs(lgtSGs[..., 1:2]), lgtSGs[..., 2:]), dim=-1)

    # ---- runner surface (idr_train.py:506,543-547,634-638,705-713,901-903) ---------------------------
    def freeze_light(self):
        self.lgtSGs.requires_grad = False

    def freeze_diffuse(self):
        for p in self.diffuse_albedo_layers.parameters():
            p.requires_grad = False

    def unfreeze_diffuse(self):
        for p in self.diffuse_albedo_layers.parameters():
            p.requires_grad = True

    def unfreeze_all(self):
        for p in self.parameters():
            p.requires_grad = True

    def freeze_all(self):
        for p in self.parameters():
            p.requires_grad = False

    def set_roughness_fake(self, state):
        self.fake_roughness = state

    def set_specular_fake(self, state):
        self.fake_specular = state

    def get_light(self):
        lgt = self.lgtSGs.clone().detach()
        if self.white_light:
            lgt = torch.cat((lgt, lgt[..., -1:], lgt[..., -1:]), dim=-1)
        if self.upper_hemi:
            lgt = self.restrict_lobes_upper(lgt)
        return lgt

    def load_light(self, path):
        assert path.endswith('.npy')
        device = self.lgtSGs.data.device
        self.lgtSGs = nn.Parameter(torch.from_numpy(np.load(path)).to(device), requires_grad=True)
        self.numLgtSGs = self.lgtSGs.data.shape[0]
        if self.lgtSGs.data.shape[1] == 7:
            self.white_light = False

    def get_base_materials(self):
        roughness = torch.sigmoid(self.roughness.clone().detach()) if not self.roughness_mlp else torch.zeros(1, 1)
        if self.fix_specular_albedo:
            spec = self.specular_reflectance
        elif not self.specular_mlp:
            spec = torch.sigmoid(self.specular_reflectance.clone().detach())
            if self.white_specular:
                spec = spec.expand((-1, 3))
        else:
            spec = torch.zeros(1, 3)
        return roughness, spec

    def get_lgtSGs(self):
        lgt = self.lgtSGs
        if self.white_light:
            lgt = torch.cat((lgt, lgt[..., -1:], lgt[..., -1:]), dim=-1)
        if self.upper_hemi:
            lgt = self.restrict_lobes_upper(lgt)
        return lgt

    @staticmethod
    def specular_remap(s):
        return 0.16 * s ** 2

    @staticmethod
    def specular_inv_remap(s):
        return (s / 0.16) ** 0.5

    def packed(self, device):
        half = ops.mlp_precision() if ops.mlp_precision() in ('f16', 'f16x3') else False
        if self._pm is None or self._pm.device != device or self._pm.half != half:
            self._pm = ops.PackedMLP(self.specs, ops.ACT_ELU, ops.HEAD_SIGMOID, self.enc, self.feature_vector_size,
                                     device, half=half)
        return self._pm

    def forward(self, points, feature_vector=None, normal=None):
        """-> dict(sg_lgtSGs [M,7], sg_specular_reflectance, sg_roughness, sg_diffuse_albedo [N,3],
        sg_blending_weights None)   (sg_envmap_material.py:357-425)."""
        p = ops._f32(points)
        feat = ops._f32(feature_vector) if (feature_vector is not None and self.feature_vector_size > 0) else None
        lins = [m for m in self.diffuse_albedo_layers if isinstance(m, nn.Linear)]
        ws = [m.weight for m in lins]
        bs = [m.bias for m in lins]
        brdf = ops.FusedMLPFn.apply(self.packed(p.device), p, None, None, feat, *ws, *bs)   # sigmoid head in-kernel
        diffuse_albedo = brdf[..., :3]
        if (not self.roughness_mlp and not self.specular_mlp and not self.fix_specular_albedo and p.is_cuda and
                self.specular_reflectance.numel() == (1 if self.white_specular else 3) and
                os.environ.get('NEFII_MATERIAL_HEAD', '1') != '0'):
            # global roughness / specular parameters (physg.conf): the whole scalar head in one launch each way
            roughness, spec = ops.MaterialHeadGlobalFn.apply(self.roughness, self.specular_reflectance, self.fake_roughness,
                                                             self.fake_specular)
            return {'sg_lgtSGs': self.get_lgtSGs(), 'sg_specular_reflectance': spec, 'sg_roughness': roughness,
                    'sg_diffuse_albedo': diffuse_albedo, 'sg_blending_weights': None}
        offset = 3
        if self.roughness_mlp:
            roughness = brdf[..., offset:offset + 1]
            offset += 1
        else:
            roughness = torch.sigmoid(self.roughness)
        if self.fix_specular_albedo:
            spec = self.specular_reflectance
        else:
            if self.specular_mlp:
                spec = brdf[..., offset:offset + 1]
                offset += 1
            else:
                spec = torch.sigmoid(self.specular_reflectance)
            if self.white_specular:
                spec = spec.expand((-1, 3))
        roughness = (1 - 0.089) * roughness + 0.089      # TINNY_ROUGHNESS :403-405
        if self.fake_roughness:
            roughness = 0 * roughness + 0.5
        if self.fake_specular:
            spec = 0 * spec + 0.5
        spec = self.specular_remap(spec)
        return {'sg_lgtSGs': self.get_lgtSGs(), 'sg_specular_reflectance': spec, 'sg_roughness': roughness,
                'sg_diffuse_albedo': diffuse_albedo, 'sg_blending_weights': None}
