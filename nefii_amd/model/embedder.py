"""Positional encoding (reference code/model/embedder.py:5-50).

The HIP kernels encode raw 3-vectors themselves (mlp_tile.h: encode_tile); this torch version is what the
trainable-geometry slow path (model/trainable_geometry.py) differentiates through, with the reference's
``get_embedder(multires)`` -> (fn, out_dim) signature."""
import torch


def get_embedder(multires):
    out_dim = 3 + 6 * multires

    def embed(x):
        parts = [x]
        for k in range(multires):
            parts += [torch.sin(x * float(2 ** k)), torch.cos(x * float(2 ** k))]
        return torch.cat(parts, -1)
    return embed, out_dim
