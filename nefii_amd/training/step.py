"""One Step-2 optimisation step with the reference's step definition (idr_train.py:750-776):
forward -> IDRLoss -> backward -> (DDP: gradient all-reduce, mean) -> idr Adam step + sg Adam step.

Optimiser construction follows idr_train.py:188-196: one Adam over implicit+rendering parameters, one over the
envmap/material parameters, both lr 5e-4.  The multi-GPU path shards the pixel batch per rank (the dataset's
contiguous patch split, scene_dataset.py:268-279) and averages gradients with ONE flat all-reduce over
RCCL/xGMI (what DistributedDataParallel does for the reference, idr_train.py:308-309), 6.6-13 MB per step."""
import torch
import torch.distributed as dist

from ..model.loss import IDRLoss


def allreduce_mean_gradients(params, world_size, group=None):
    """Average .grad of `params` across ranks with a single flat all-reduce (no-op for world_size 1)."""
    if world_size <= 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return flat.numel() * flat.element_size()


class TrainStep:
    def __init__(self, model, loss_conf, idr_lr=5e-4, sg_lr=5e-4, world_size=1):
        self.model = model
        self.loss = IDRLoss(**loss_conf)
        self.world_size = world_size
        self.idr_optimizer = torch.optim.Adam(list(model.implicit_network.parameters()) +
                                              list(model.rendering_network.parameters()), lr=idr_lr)
        self.sg_optimizer = torch.optim.Adam(model.envmap_material_network.parameters(), lr=sg_lr)
        self.trainable = [p for p in model.parameters() if p.requires_grad]

    def __call__(self, model_input, ground_truth):
        out = self.model(model_input)
        lo = self.loss(out, ground_truth)
        self.idr_optimizer.zero_grad()
        self.sg_optimizer.zero_grad()
        lo['loss'].backward()
        allreduce_mean_gradients(self.trainable, self.world_size)
        self.idr_optimizer.step()
        self.sg_optimizer.step()
        return out, lo
