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
    if flat.is_cuda and dist.get_backend(group) == 'gloo':
        # CPU-staged path for smoke-testing the multi-process logic on a box without RCCL peers
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return flat.numel() * flat.element_size()


class TrainStep:
    def __init__(self, model, loss_conf, idr_lr=5e-4, sg_lr=5e-4, world_size=1, secondary_train_interval=0,
                 secondary_batch_size=1024, num_rays=1):
        self.model = model
        self.loss = IDRLoss(**loss_conf)
        self.world_size = world_size
        # secondary-point consistency step (idr_train.py:44,788,804-852): every `interval` iterations, on the first
        # secondary_batch_size // world masked secondary hits, each replicated num_rays times
        self.secondary_train_interval = secondary_train_interval
        self.secondary_batch_size = secondary_batch_size // max(world_size, 1)
        self.num_rays = max(num_rays, 1)
        self.cur_iter = 0
        # same Adam as the reference (idr_train.py:188-196); `fused` only selects torch's single-kernel
        # implementation of the identical update when the parameters live on the GPU
        fused = next(model.parameters()).is_cuda
        self.idr_optimizer = torch.optim.Adam(list(model.implicit_network.parameters()) +
                                              list(model.rendering_network.parameters()), lr=idr_lr, fused=fused)
        self.sg_optimizer = torch.optim.Adam(model.envmap_material_network.parameters(), lr=sg_lr, fused=fused)
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
        if self.secondary_train_interval > 0 and self.cur_iter % self.secondary_train_interval == 0:
            self.train_with_secondary(out)
        self.cur_iter += 1
        return out, lo

    def train_with_secondary(self, model_outputs):
        """L1(sg_rgb, idr_rgb) at secondary hit points, seen from the direction they were hit from
        (idr_train.py:804-852): ties the material/light decomposition to the radiance field where the camera
        never looks."""
        pts, mask, dirs = (model_outputs.get(k) for k in ('secondary_points', 'secondary_mask', 'secondary_dir'))
        if pts is None or mask is None or dirs is None:
            return None
        m = mask.reshape(-1)
        idx = torch.nonzero(m).flatten()[:self.secondary_batch_size]
        if idx.numel() == 0:
            return None
        p = pts.detach().reshape(-1, 3).index_select(0, idx)
        d = dirs.detach().reshape(-1, 3).index_select(0, idx)
        n = p.shape[0]
        ret = self.model({'points': p.unsqueeze(1).expand(n, self.num_rays, 3),
                          'ray_dirs': d.unsqueeze(1).expand(n, self.num_rays, 3)}, with_point=True)
        loss = torch.nn.functional.l1_loss(ret['sg_rgb_values'], ret['idr_rgb_values'])
        self.idr_optimizer.zero_grad()
        self.sg_optimizer.zero_grad()
        loss.backward()
        allreduce_mean_gradients(self.trainable, self.world_size)
        self.idr_optimizer.step()
        self.sg_optimizer.step()
        return loss
