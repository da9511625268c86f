"""render_with_sg with the reference's signature (code/model/sg_render.py:164-295), on the fused HIP kernel.

Supported: one base material (K = 1) with global roughness [1,1] and specular [1,3] and no blending weights -
what physg.conf-style models pass.  (conf.conf's per-point roughness asserts inside the reference's
render_with_sg too: sg_render.py:177.)"""
import torch

from .. import ops

TINY_NUMBER = 1e-6


def render_with_sg(lgtSGs, specular_reflectance, roughness, diffuse_albedo, normal, viewdirs, blending_weights=None,
                   diffuse_rgb=None):
    """lgtSGs [M,7]; specular_reflectance [1,3]; roughness [1,1]; diffuse_albedo / normal / viewdirs [...,3]
    -> dict(sg_rgb, sg_specular_rgb, sg_diffuse_rgb, sg_diffuse_albedo) [...,3]."""
    K = specular_reflectance.shape[0]
    assert K == roughness.shape[0]
    if K != 1 or blending_weights is not None or diffuse_rgb is not None:
        raise NotImplementedError('render_with_sg: only K=1 without blending weights (all shipped confs)')
    shape = normal.shape[:-1]
    n = normal.reshape(-1, 3)
    v = viewdirs.reshape(-1, 3)
    a = diffuse_albedo.reshape(-1, 3)
    rgb, spec, diff = ops.SGRenderFn.apply(lgtSGs, specular_reflectance, roughness, a, n, v)
    return {'sg_rgb': rgb.reshape(*shape, 3), 'sg_specular_rgb': spec.reshape(*shape, 3),
            'sg_diffuse_rgb': diff.reshape(*shape, 3), 'sg_diffuse_albedo': diffuse_albedo}
