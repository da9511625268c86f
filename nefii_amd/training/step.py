"""One Step-2 optimisation step with the reference's step definition (idr_train.py:750-776):
forward -> IDRLoss -> backward -> (DDP: gradient all-reduce, mean) -> idr Adam step + sg Adam step.

Optimiser construction follows idr_train.py:188-196: one Adam over implicit+rendering parameters, one over the
envmap/material parameters, both lr 5e-4.  The multi-GPU path shards the pixel batch per rank (the dataset's
contiguous patch split, scene_dataset.py:268-279) and averages gradients with ONE flat all-reduce over
RCCL/xGMI (what DistributedDataParallel does for the reference, idr_train.py:308-309), 6.6-13 MB per step."""
import contextlib
import os

import torch
import torch.distributed as dist

from .. import ops
from ..model.loss import IDRLoss


def _all_reduce_sum(flat, group=None):
    """SUM all-reduce of one flat tensor; GPU tensors on a gloo group (multi-process smoke tests on a box without RCCL
    peers) are staged through the host."""
    if flat.is_cuda and dist.get_backend(group) == 'gloo':
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_parameters(module, src=0, group=None):
    """Every rank takes rank `src`'s parameters and buffers (what DistributedDataParallel does when the reference wraps
    the model, idr_train.py:308-309): ranks build their networks from their own unseeded RNG, and only gradients are
    exchanged afterwards.  One flat broadcast per dtype."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return 0
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.numel() > 0]
    total = 0
    for dtype in sorted({t.dtype for t in tensors}, key=str):
        part = [t for t in tensors if t.dtype == dtype]
        flat = torch.cat([t.detach().reshape(-1) for t in part])
        if flat.is_cuda and dist.get_backend(group) == 'gloo':
            host = flat.cpu()
            dist.broadcast(host, src=src, group=group)
            flat.copy_(host)
        else:
            dist.broadcast(flat, src=src, group=group)
        off = 0
        with torch.no_grad():
            for t in part:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        total += flat.numel() * flat.element_size()
    return total


def allreduce_mean_gradients(params, world_size, group=None, flags=None):
    """Average .grad of `params` across ranks with ONE flat all-reduce (no-op for world_size 1).

    The buffer has a FIXED layout: one slot per parameter of `params` whether or not it received a gradient on this
    rank (zeros then), followed by the caller's `flags` (a 1-D float tensor, summed over the ranks and returned).
    Every rank therefore sends the same number of bytes and always enters the collective - a rank whose pixel slice
    contains no hit has no gradient at all for the radiance and material networks, and must neither skip the
    all-reduce nor send a shorter buffer.  Semantics of DistributedDataParallel(find_unused_parameters=True) under
    the reference's zero-filling optimizer.zero_grad() (idr_train.py:309,760-761): the mean is over ALL ranks, a rank
    without a gradient contributes zeros, and afterwards every parameter holds a gradient tensor (Adam on an
    all-zero gradient with zero moments is a no-op).  No host synchronisation.
    Returns (bytes reduced, summed flags or None)."""
    if world_size <= 1:
        return 0, flags
    params = list(params)
    if not params:
        return 0, flags
    dev, dtype = params[0].device, params[0].dtype
    parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dtype) for p in params]
    if flags is not None:
        parts.append(flags.reshape(-1).to(device=dev, dtype=dtype))
    flat = torch.cat(parts)
    _all_reduce_sum(flat, group)
    n_par = sum(p.numel() for p in params)
    flat[:n_par].div_(world_size)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    out_flags = flat[n_par:].clone() if flags is not None else None
    return flat.numel() * flat.element_size(), out_flags


class FlatGrads:
    """Every trainable parameter's .grad as a VIEW into one persistent flat buffer (+ a tail of flag slots): the multi-rank
    step zeroes the buffer with one fill, backward accumulates into the views in place, and the gradient exchange is ONE
    all-reduce of the buffer itself - no torch.cat to build it and no per-parameter copy back (round 3: ~3 launches per
    parameter, ~80 per step).  Fixed layout, as allreduce_mean_gradients: a parameter that received no gradient on this rank
    contributes its zeros, every rank sends the same bytes and always enters the collective."""

    def __init__(self, params, n_flags=1):
        self.params = list(params)
        dev = self.params[0].device
        self.n_par = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.n_par + n_flags, device=dev, dtype=torch.float32)
        self.flags = self.flat[self.n_par:]
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def zero(self):
        """the step's zero_grad(): one fill, and every parameter's .grad (re-)pointed at its view"""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def adopt(self):
        """gradients that autograd (or a captured graph) left in tensors of their own: copied into the views once"""
        for p, v in zip(self.params, self.views):
            if p.grad is not None and p.grad is not v and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v

    def allreduce_mean(self, world_size, group=None):
        _all_reduce_sum(self.flat, group)
        self.flat[:self.n_par].mul_(1.0 / world_size)
        return self.flat.numel() * 4


def _detached(d):
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()}


class _StepGraph:
    """One captured hipGraph of the step's tail for a fixed padded hit count: shading of the compacted hit rays,
    IDRLoss, backward and - single process - both Adam updates.  Inputs live in static buffers."""

    def __init__(self, step, ctx, idx_pad, dst_pad, ground_truth):
        self.static_ctx = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in ctx.items() if k != 'pre'}
        pre = ctx['pre']
        self.static_ctx['pre'] = None if pre is None else tuple(None if t is None else t.clone() for t in pre)
        self.idx, self.dst = idx_pad.clone(), dst_pad.clone()
        self.gt = {k: v.clone() for k, v in ground_truth.items() if torch.is_tensor(v)}
        self.graph = torch.cuda.CUDAGraph()
        # one process: gradients are allocated inside the capture.  Several ranks: they are views of the step's flat
        # buffer - the capture zeroes it (one fill) and backward accumulates in place; the all-reduce and both Adam updates
        # follow the replay eagerly (3 launches + 2), or run INSIDE the graph with NEFII_GRAPH_COLLECTIVES=1 on an RCCL
        # group (capture of RCCL kernels: never run on hardware by this repo - off by default)
        in_graph = step.world_size <= 1 or step.graph_collectives
        if step._flat is None:
            step._zero_grads(set_to_none=True)
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):   # RCCL's watchdog thread stays legal
            self.out = step.model.shade_tail(self.static_ctx, self.idx, self.dst)
            self.lo = step.loss(self.out, self.gt)
            if step._flat is not None:
                step._zero_grads()
            self.lo['loss'].backward()
            if in_graph:
                step._update(self.lo['loss'])
        self.updates_in_graph = in_graph
        self.grads = [(p, p.grad) for p in step.trainable]      # the tensors every replay writes its gradients to
        # keep the static result tensors, not the autograd graph behind them: AccumulateGrad nodes that outlive their
        # iteration are re-used by the next backward on THEIR stream, which breaks the next capture
        self.out = _detached(self.out)
        self.lo = _detached(self.lo)

    def load(self, ctx, idx_pad, dst_pad, ground_truth):
        dsts, srcs = [self.idx, self.dst], [idx_pad, dst_pad]
        for k, v in ctx.items():
            if k == 'pre' and v is not None:
                for d, t in zip(self.static_ctx['pre'], v):
                    if t is not None:
                        dsts.append(d), srcs.append(t)
            elif torch.is_tensor(v):
                dsts.append(self.static_ctx[k]), srcs.append(v)
        for k, v in self.gt.items():
            dsts.append(v), srcs.append(ground_truth[k])
        # one multi-tensor launch per dtype instead of a dozen copies
        by_type = {}
        for d, t in zip(dsts, srcs):
            if d.dtype == t.dtype and d.shape == t.shape and d.is_contiguous() and t.is_contiguous():
                by_type.setdefault(d.dtype, ([], []))
                by_type[d.dtype][0].append(d), by_type[d.dtype][1].append(t)
            else:
                d.copy_(t)
        for ds, ts in by_type.values():
            torch._foreach_copy_(ds, ts)


def _hit_index_ahead(ctx):
    """On the trace stream, behind the trace: the hit rays' indices (padded to all rays) and their count in pinned host
    memory - so that the step that consumes this trace needs no torch.nonzero, whose host sync would sit on the CALLER'S
    stream behind the previous step's tail and keep the host from enqueueing one step while the GPU runs the other."""
    mask = ctx['network_object_mask']
    n_all = mask.shape[0]
    ctx['hit_idx_all'] = torch.nonzero_static(mask, size=n_all, fill_value=n_all).flatten()
    # the same list with its padding pointing at a ray that exists: what the graph step gathers FROM (it scatters TO the list
    # above, whose padding names the scratch row n_all)
    ctx['hit_idx_src'] = ctx['hit_idx_all'].clamp(max=n_all - 1)
    host = torch.empty(1, dtype=torch.int64, pin_memory=True)
    host.copy_(mask.sum(dtype=torch.int64).reshape(1), non_blocking=True)
    ctx['hit_count_host'] = host


def _hit_index(ctx):
    """Indices of the hit rays: what _hit_index_ahead prepared (valid once the trace's event is done), else torch.nonzero."""
    if 'hit_idx_all' in ctx:
        return ctx['hit_idx_all'][:int(ctx['hit_count_host'].item())]
    return torch.nonzero(ctx['network_object_mask']).flatten()


def _scheduler_step(sched):
    """MultiStepLR.step() - on the iterations that are no milestone as two counter increments: the learning rate does not
    change there, but torch's step() recomputes it, writes it back (graph mode: a fill_ launch per device-resident lr) and
    clones it into _last_lr (another launch per group) - six tiny launches on the caller's stream and 0.045 ms of host time
    per training step.  State (last_epoch, _step_count, _last_lr, the groups' lr) is what step() would have left."""
    nxt = sched.last_epoch + 1
    if nxt in sched.milestones:
        sched.step()
    else:
        sched.last_epoch = nxt
        sched._step_count += 1


_TRACE_POOLS = {}


def _trace_pool(dev, n):
    """The trace streams are the PROCESS's, not the TrainStep's: HIP multiplexes its streams onto a handful of hardware
    queues, and every further TrainStep that made its own streams found them sharing queues with the ones the step's tail
    and the tracer's round groups run on - config 1 measured 0.92 ms per step with the first TrainStep of a process and
    1.19 ms with every later one (tools/experiments/nested_cfg1.py: not the garbage collector, not the allocator)."""
    pool = _TRACE_POOLS.setdefault(dev, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


class TrainStep:
    def __init__(self, model, loss_conf, idr_lr=5e-4, sg_lr=5e-4, world_size=1, secondary_train_interval=0,
                 secondary_batch_size=1024, num_rays=1, graph=False, graph_bucket=256, graph_after=3,
                 idr_sched_milestones=(), idr_sched_factor=0.0, sg_sched_milestones=(), sg_sched_factor=0.0,
                 alpha_milestones=(), alpha_factor=0.0, roughness_warmup=-1, specular_warmup=-1, start_iter=0,
                 min_sdf_every=None):
        """graph=True: after `graph_after` eager iterations the part of the step behind the tracer - whose launch count
        (~170 small kernels) rather than its GPU time bounds it - replays as a captured hipGraph.  The hit count varies
        from batch to batch, so the compacted index list is padded to a multiple of `graph_bucket` (padding rows
        scatter into a scratch output row and receive zero gradient: same loss and gradients as the eager step up to
        summation order) and one graph is kept per padded size.  Only the closed-form `sg` shading qualifies (the
        Monte-Carlo path traces secondary rays, with data-dependent shapes of its own).

        The per-iteration hooks of the reference's loop (idr_train.py:692-713,799-802) are part of the step: the mask
        loss's alpha doubles (x alpha_factor) at alpha_milestones, the material network reports fake roughness /
        specular while cur_iter < roughness_warmup / specular_warmup, and both MultiStepLR schedulers
        (train.{idr,sg}_sched_milestones / _factor, idr_train.py:190-198) advance once per iteration."""
        self.model = model
        self.loss = IDRLoss(**loss_conf)
        self.world_size = world_size
        # The tracer's training-mode min-SDF search (ray_tracing.py:309-337: 100 samples per ray that misses - 57 % of config
        # 3's single-pass evaluations, 76 % of config 2's) fills `points` / `sdf_output` of the MISS rays, and under frozen
        # geometry those reach nothing but the VALUE of mask_loss: no gradient - and the
        # reference reads the loss value only in its NaN check and in the line it prints every 50 iterations
        # (idr_train.py:754,784).  min_sdf_every = E > 1 runs the search on the iterations with cur_iter % E == 0 only (the
        # runner passes its logging period): the logged losses are the reference's, the gradients - hence parameters and
        # optimizer state - are those of the every-iteration schedule (test_min_sdf_on_reporting_iterations_only: as close as
        # two runs of one schedule are to each other - the light's gradient is summed with float atomics), the losses returned on
        # the other iterations carry a mask_loss computed without the search (finite whenever the true one is).  The
        # search's uniform draw is made on every iteration either way.  Default (None): NEFII_MIN_SDF_EVERY, else 1 - every
        # iteration, the reference's schedule; trainable geometry always runs it.
        self.coarse_events = []         # (iteration, 'recalibrated' | 'disabled', observed, bound): the coarse bound's audit
        self.retraced_steps = 0         # batches traced again because the bound their enqueued trace assumed did not hold
        if min_sdf_every is None:
            min_sdf_every = int(os.environ.get('NEFII_MIN_SDF_EVERY', '1'))
        self.min_sdf_every = max(1, int(min_sdf_every))
        # secondary-point consistency step (idr_train.py:44,788,804-852): every `interval` iterations, on the first
        # secondary_batch_size // world masked secondary hits, each replicated num_rays times
        self.secondary_train_interval = secondary_train_interval
        self.secondary_batch_size = secondary_batch_size // max(world_size, 1)
        self.num_rays = max(num_rays, 1)
        self.cur_iter = int(start_iter)
        self.alpha_milestones, self.alpha_factor = tuple(alpha_milestones), alpha_factor
        self.roughness_warmup, self.specular_warmup = int(roughness_warmup), int(specular_warmup)
        for acc in self.alpha_milestones:          # resuming past a milestone (idr_train.py:325-327)
            if self.cur_iter > acc:
                self.loss.alpha = self.loss.alpha * self.alpha_factor
        # same Adam as the reference (idr_train.py:188-196); `fused` only selects torch's single-kernel
        # implementation of the identical update when the parameters live on the GPU
        fused = next(model.parameters()).is_cuda
        self._fused = fused
        # flag and unit scale of the fused NaN guard (_update): allocated here, outside any graph capture - with
        # graph_after = 0 the first _update runs inside torch.cuda.graph and would take them from the graph's private pool
        _dev = next(model.parameters()).device
        self._found = torch.zeros((), device=_dev, dtype=torch.float32)
        self._one = torch.ones((), device=_dev, dtype=torch.float32)
        self.graph = bool(graph) and fused and getattr(model, 'render_type', None) == 'sg'
        self.graph_bucket, self.graph_after = int(graph_bucket), int(graph_after)
        self.graph_collectives = (world_size > 1 and os.environ.get('NEFII_GRAPH_COLLECTIVES', '0') == '1' and
                                  dist.is_initialized() and dist.get_backend() == 'nccl')
        self._graphs = {}
        self._eager_steps = 0     # captures need an initialised optimiser state: the first steps run eagerly
        self._prefetch, self._trace_stream, self._trace_pool = [], None, []   # traces enqueued ahead: (input, ctx, event, checks)
        # the SDF value/gradient pass at the traced points: on the trace stream it sits on the step's critical path; in the
        # tail (default) it runs beside the next trace (config 2: 5.2 vs 5.45 ms per step)
        # - for batches traced one at a time.  Small batches traced in groups (trace_group_for) are bound by the serial chain of
        # the tail on the caller's stream: their pass runs on the trace stream, with four groups in flight on four streams
        # (config 1: 1.40 -> 1.08 ms per step together with the hit index computed ahead, _hit_index_ahead).
        # NEFII_SURFACE_IN_TAIL = 0 / 1 overrides for both
        env = os.environ.get('NEFII_SURFACE_IN_TAIL')
        self.surface_in_tail = True if env is None else env == '1'
        self.surface_in_tail_grouped = False if env is None else env == '1'
        kw = dict(fused=fused, capturable=True) if self.graph else dict(fused=fused)
        if self.graph:      # a captured Adam reads its learning rate from device memory: the schedulers fill it in place
            dev = next(model.parameters()).device
            idr_lr, sg_lr = torch.tensor(float(idr_lr), device=dev), torch.tensor(float(sg_lr), device=dev)
        self.idr_optimizer = torch.optim.Adam(list(model.implicit_network.parameters()) +
                                              list(model.rendering_network.parameters()), lr=idr_lr, **kw)
        self.sg_optimizer = torch.optim.Adam(model.envmap_material_network.parameters(), lr=sg_lr, **kw)
        self.idr_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.idr_optimizer, list(idr_sched_milestones),
                                                                  gamma=idr_sched_factor)
        self.sg_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.sg_optimizer, list(sg_sched_milestones),
                                                                 gamma=sg_sched_factor)
        self.trainable = [p for p in model.parameters() if p.requires_grad]
        # several ranks: gradients live in one flat buffer that is all-reduced as it stands (FlatGrads)
        self._flat = FlatGrads(self.trainable) if world_size > 1 and self.trainable else None
        # physg.conf weights the radiance colour with 0 and IDRLoss detaches it: the radiance network then runs without
        # autograd, weight norm and repacking (its weights never change).  Not with the secondary-consistency step, whose own
        # loss reads the radiance colour
        rn = getattr(model, 'rendering_network', None)
        if rn is not None and hasattr(rn, 'outputs_detached'):
            # ... and only for the closed-form 'sg' render type: the Monte-Carlo render types also reach the radiance
            # network through the indirect light at secondary hits (path_tracing_render.py: rendering_network(hp, hn, hv,
            # feats) with autograd on, as the reference's get_visibility_and_indirect_light) - detached there, an MC conf
            # with idr_rgb_weight = 0 would silently stop training it
            rn.outputs_detached = bool(self.loss.idr_rgb_weight == 0 and secondary_train_interval == 0
                                       and getattr(model, 'render_type', 'sg') == 'sg')
        # steps whose loss was not finite on some rank: their gradients were zeroed on every rank before Adam ran
        # (device counter: the runner reads it at its logging points only)
        self.nonfinite_steps = torch.zeros((), device=next(model.parameters()).device, dtype=torch.float32)
        if world_size > 1:
            # the reference's DDP wrap broadcasts rank 0's parameters and buffers at construction (idr_train.py:308-309);
            # without it every rank would train the replica its own RNG initialised
            broadcast_parameters(model)

    def _zero_grads(self, set_to_none=None):
        if self._flat is not None:
            self._flat.zero()
            return
        if set_to_none is None:
            self.idr_optimizer.zero_grad()
            self.sg_optimizer.zero_grad()
        else:
            self.idr_optimizer.zero_grad(set_to_none=set_to_none)
            self.sg_optimizer.zero_grad(set_to_none=set_to_none)

    def _update(self, loss):
        """Gradient exchange + both Adam updates, with the reference's NaN check (idr_train.py:754-757: before
        backward / step) as a device-side guard: when the loss - or a gradient - is not finite on ANY rank, every rank
        zeroes its gradients and both optimizers skip the step: parameters, moments and step counters are those of the
        last good iteration, so the emergency checkpoint the runner writes is the pre-NaN state.  The flag travels in
        the gradient all-reduce: one collective, no host sync.
        After a cancelled step `.grad` is UNDEFINED on the fused (GPU) path - the non-finite values stay where backward put
        them (several ranks: the all-reduce spreads them) and only the optimizers look at the flag; anything that reads
        gradients between steps (a gradient-norm log) must check `nonfinite_steps` first.  The check itself is torch's amp
        multi-tensor kernel (`torch._amp_foreach_non_finite_check_and_unscale_`, what GradScaler.unscale_ runs) and the skip is
        the fused Adam's `found_inf` input: private interfaces, pinned by
        test_nonfinite_step_is_skipped_not_run_on_zero_gradients (eager and graph) against a torch upgrade."""
        grads = [p.grad for p in self.trainable if p.grad is not None]
        if self._fused:
            # one multi-tensor kernel (the one torch.amp.GradScaler unscales with; scale 1.0 leaves every value as it is)
            # instead of isfinite(loss) + foreach_norm + isfinite(norms): 2 launches instead of ~25 in the step's tail
            self._found.zero_()
            torch._amp_foreach_non_finite_check_and_unscale_(grads + [loss.detach().reshape(1).clone()], self._found, self._one)
            bad = self._found.reshape(1)
        else:
            bad = (~torch.isfinite(loss.detach())).reshape(1).to(torch.float32)
            if grads:   # a finite loss can still come with a non-finite gradient (0 x inf in some backward): same treatment
                norms = torch.stack(torch._foreach_norm(grads))
                bad = bad + (~torch.isfinite(norms).all()).reshape(1).to(torch.float32)
        if self.world_size > 1:
            if self._flat is not None:
                self._flat.adopt()
                self._flat.flags.copy_(bad.reshape(-1).to(torch.float32))
                self._flat.allreduce_mean(self.world_size)
                bad = self._flat.flags.clone()
            else:
                _, bad = allreduce_mean_gradients(self.trainable, self.world_size, flags=bad)
        if self._fused:             # the flag stays a float 0 / 1 on the device: no compare / cast launches
            bad = bad.reshape(()).clamp(max=1.0) if self.world_size > 1 else bad.reshape(())
            self.nonfinite_steps += bad
        else:
            bad = bad.reshape(()) > 0
            for p in self.trainable:        # (the fused Adam skips by flag: no need to touch the gradients there)
                if p.grad is not None:
                    p.grad.masked_fill_(bad, 0.0)
            self.nonfinite_steps += bad.to(self.nonfinite_steps.dtype)
        # ... and the optimizers SKIP the step (zero gradients alone would still move the parameters by the first
        # moment, decay both moments and advance the step counters).  On the GPU the fused Adam takes the flag the way
        # torch.amp.GradScaler hands it over (optimizer.found_inf: the kernel leaves parameters and moments alone and
        # the step counter is taken back) - on the device, capturable; the CPU implementation (multi-process gloo tests)
        # has no such input, there the flag is read.
        if self._fused:
            self.idr_optimizer.found_inf = self.sg_optimizer.found_inf = bad
            self.idr_optimizer.step()
            self.sg_optimizer.step()
        elif not bool(bad):
            self.idr_optimizer.step()
            self.sg_optimizer.step()

    def retensor_lr(self):
        """After optimizer.load_state_dict, which installs the CHECKPOINT's param_groups and state: a reference checkpoint
        (or one saved by an eager run) carries python-float learning rates, capturable=False / fused=None and 'step'
        counters on the CPU.  Graph mode needs them as device tensors that the captured Adam reads and the schedulers
        fill in place, with capturable=True - otherwise the capture at iteration `graph_after` raises."""
        if not self.graph:
            return
        for opt in (self.idr_optimizer, self.sg_optimizer):
            dev = opt.param_groups[0]['params'][0].device
            for g in opt.param_groups:
                if not torch.is_tensor(g['lr']):
                    g['lr'] = torch.tensor(float(g['lr']), device=dev)
                g['capturable'] = True
                g['fused'] = True
                g['foreach'] = None
            for st in opt.state.values():
                if 'step' in st:
                    st['step'] = torch.as_tensor(st['step'], dtype=torch.float32).to(dev)
        self._graphs.clear()          # captured graphs hold the old tensors

    @staticmethod
    def portable_state_dict(opt):
        """optimizer.state_dict() with python-float learning rates (what the reference's runner saves and loads)."""
        sd = opt.state_dict()
        for g in sd['param_groups']:
            if torch.is_tensor(g.get('lr')):
                g['lr'] = float(g['lr'])
        return sd

    @contextlib.contextmanager
    def _min_sdf_schedule(self, *iters):
        """The tracer's schedule while the trace(s) of the given iteration(s) are enqueued (several: batches traced as one
        call).  The two switches live on the model's ray tracer, which other callers share (a second TrainStep, a direct
        training-mode model(...) call, bench.py's side measurements): they are set for the enqueue only and put back."""
        if self.min_sdf_every <= 1 or not getattr(self.model, 'state_freeze_geo', False):
            yield
            return
        rt = self.model.ray_tracer
        old = (rt.skip_min_sdf_search, rt.draw_when_skipped)
        rt.skip_min_sdf_search = not any(i % self.min_sdf_every == 0 for i in iters)
        rt.draw_when_skipped = True
        try:
            yield
        finally:
            rt.skip_min_sdf_search, rt.draw_when_skipped = old

    def _pre_iteration(self):
        """idr_train.py:692-713, in the reference's order."""
        if self.cur_iter in self.alpha_milestones:
            self.loss.alpha = self.loss.alpha * self.alpha_factor
        mat = self.model.envmap_material_network
        if self.cur_iter < self.roughness_warmup:
            mat.set_roughness_fake(True)
        elif self.cur_iter == self.roughness_warmup:
            mat.set_roughness_fake(False)
        if self.cur_iter < self.specular_warmup:
            mat.set_specular_fake(True)
        elif self.cur_iter == self.specular_warmup:
            mat.set_specular_fake(False)

    def _post_iteration(self):
        """idr_train.py:799-802."""
        self.cur_iter += 1
        _scheduler_step(self.idr_scheduler)
        _scheduler_step(self.sg_scheduler)

    # ---- trace of the NEXT batch beside the tail of this one -------------------------------------------------------
    # With frozen geometry the tracer's result does not depend on any trainable parameter, so the next batch can be
    # traced while this batch's shading / backward / Adam runs: the tail's small kernels (24-128 workgroups) fill the
    # CUs that the tracer's latency-bound rounds (fewer tiles than CUs) leave idle.  Same arithmetic, same order of the
    # tracer's random draws; only the schedule changes.
    def trace_group_for(self, model_input):
        """Batches traced per tracer call when their rays are enqueued ahead (prefetch_group).  Tiny batches are traced
        several at a time: a lone trace of a few hundred rays is all launch and tile latency, and one call of several
        batches costs hardly more than one of one.  Config 1 (512 rays), ms per step by group size, round 5: 3: 0.89, 4: 0.72,
        6: 0.63, 8: 0.61, 12: 0.63 - eight (4096 rays per call; before the coarse pass and the staged min-SDF search reached
        such calls, three was the optimum: 1.95 -> 1.63).  Config 2 (4096 rays), on the round's final tree: 1: 2.27, 3: 2.18, 4:
        2.04, 6: 2.01 - four (16 384 rays per call: a call stays below RayTracing.TIER_MIN_RAYS, so grouping never changes which
        arithmetic a batch is traced with).  Bigger batches fill the chip on their own: 1.  NEFII_TRACE_GROUP overrides."""
        env = os.environ.get('NEFII_TRACE_GROUP')
        if env:
            return max(1, int(env))
        uv = model_input['uv']
        if uv.shape[0] != 1:
            return 1
        n_rays = uv.shape[1] * (uv.shape[2] if uv.dim() == 4 else 1)
        return 8 if n_rays <= 1024 else (4 if n_rays <= 4096 else 1)

    def preferred_lookahead(self, model_input):
        """Upcoming batches a caller should hand to __call__ (next_input): three traces in flight for big batches; for
        grouped traces four groups minus one: a group is enqueued while three whole traced groups are in flight or waiting (their
        rounds are all latency; config 1: 1.25 ms per step with two groups, 1.08 with four) - 31 batches for config 1's groups
        of eight (frozen geometry: a trace depends on no trained parameter, however far ahead it is made)."""
        g = self.trace_group_for(model_input)
        return 3 if g <= 1 else 4 * g - 1

    def _trace_stream_next(self, after, grouped=False):
        if self._trace_stream is None:
            n = max(1, int(os.environ.get('NEFII_TRACE_STREAMS', '4' if grouped else '3')))
            self._trace_pool = list(_trace_pool(torch.cuda.current_device(), n))
        self._trace_pool.append(self._trace_pool.pop(0))
        self._trace_stream = self._trace_pool[0]
        self._trace_stream.wait_event(after if after is not None else torch.cuda.current_stream().record_event())
        return self._trace_stream

    def prefetch_group(self, inputs, after=None):
        """prefetch_trace for several upcoming batches as ONE tracer call (IDRNetwork.trace_points_group)."""
        m = self.model
        if len(inputs) == 1:
            return self.prefetch_trace(inputs[0], after)
        if not (m.training and getattr(m, 'state_freeze_geo', False) and next(m.parameters()).is_cuda):
            return
        st = self._trace_stream_next(after, grouped=True)
        checks = []
        with torch.cuda.stream(st):
            m.ray_tracer.deferred_checks = checks
            m.ray_tracer.concurrent = True
            try:
                ctxs = m.trace_points_group(inputs)
                for c in ctxs:
                    if not self.surface_in_tail_grouped:
                        m.attach_surface(c)
                    _hit_index_ahead(c)
            finally:
                m.ray_tracer.deferred_checks = None
                m.ray_tracer.concurrent = False
            ev = st.record_event()
        grp = {'done': False, 'ctxs': ctxs}
        for inp, c in zip(inputs, ctxs):
            self._prefetch.append((inp, c, ev, checks, grp))

    def prefetch_trace(self, model_input, after=None):
        """after: event on the caller's stream behind which the inputs are ready (default: now).  Must not be an event
        behind this step's tail - the trace would wait for exactly what it is meant to run beside."""
        m = self.model
        if not (m.training and getattr(m, 'state_freeze_geo', False) and next(m.parameters()).is_cuda):
            return
        # consecutive traces rotate over the streams: the trace enqueued now starts beside the one(s) still running -
        # its dense rounds fill what the others' latency-bound rounds leave idle (config 2, 1 / 2 streams: 5.26 / 5.06 ms in
        # round 1; with the coarse pass 2 / 3 / 4 streams and batches of lookahead: 3.68 / 3.34 / 3.94 ms)
        self._trace_stream_next(after)
        checks = []
        with torch.cuda.stream(self._trace_stream):
            m.ray_tracer.deferred_checks = checks       # no host sync inside the trace: its round-prefix check waits
            # traces that run beside other work are throughput-bound, not latency-bound: 3 speculative bisection levels
            # (7 nodes per ray and round) instead of the 5 (31 nodes) a lone small batch prefers - fewer wasted queries,
            # a few more rounds, bit-identical result (config 2: 4.78 vs 5.02 ms per step); RayTracing.auto_levels decides
            m.ray_tracer.concurrent = True
            try:
                ctx = m.trace_points(model_input) if self.surface_in_tail else m.trace_head(model_input)
                _hit_index_ahead(ctx)
            finally:
                m.ray_tracer.deferred_checks = None
                m.ray_tracer.concurrent = False
            ev = self._trace_stream.record_event()
        self._prefetch.append((model_input, ctx, ev, checks, None))

    def _n_prefetched(self, model_input):
        return sum(1 for pf in self._prefetch if pf[0] is model_input)

    def _take_prefetched(self, model_input):
        if not self._prefetch or self._prefetch[0][0] is not model_input:
            self._prefetch = []         # the caller changed its mind about the next batch
            return None
        pf = self._prefetch.pop(0)      # oldest first: enqueue order
        _, ctx, ev, checks, grp = pf
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        ev.synchronize()
        net = self.model.implicit_network
        seen = len(net.coarse_audit_events)
        if grp is None or not grp['done']:
            for chk in checks:
                more = chk()
                if more is not None:        # the guessed round prefix was too short: the check ran the remaining rounds
                    group = [ctx] if grp is None else grp['ctxs']
                    S = more[0].shape[0] // len(group)
                    for g, c in enumerate(group):
                        c['points'], c['network_object_mask'] = more[0][g * S:(g + 1) * S], more[1][g * S:(g + 1) * S]
                        c.pop('pre', None)
                        c.pop('hit_idx_all', None)
                        c.pop('hit_idx_src', None)
            if grp is not None:
                grp['done'] = True
        # the online audit of the tracer's coarse bound reports with these checks (ImplicitNetwork.note_coarse_audit).  A
        # bound that did NOT hold for this trace means one of its unrefined samples may have decided differently: the trace -
        # and every trace already enqueued under the same bound - is dropped and the batch traced again (the coarse pass is
        # off for these weights by now).  The re-trace draws its min-SDF uniforms anew.  Recorded for the runner's log.
        fresh = net.coarse_audit_events[seen:]
        for kind, observed, bound in fresh:
            self.coarse_events.append((self.cur_iter, kind, observed, bound))
        if any(kind in ('disabled', 'lipschitz_disabled') for kind, _, _ in fresh):
            self._prefetch = []
            self.retraced_steps += 1
            return None
        for v in list(ctx.values()) + list(ctx.get('pre') or ()):
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)          # allocated on the trace stream, consumed here
        return ctx if 'pre' in ctx else self.model.attach_surface(ctx)

    def _graph_step(self, model_input, ground_truth, ctx=None):
        if ctx is None:
            with self._min_sdf_schedule(self.cur_iter):
                ctx = self.model.trace_head(model_input)
        idx = _hit_index(ctx)      # no host sync for a trace enqueued ahead; torch.nonzero (the step's one sync) otherwise
        n_hit, n_all = idx.numel(), ctx['points'].shape[0]
        if n_hit == 0:
            return None
        P = -(-n_hit // self.graph_bucket) * self.graph_bucket
        pad = P - n_hit
        if 'hit_idx_all' in ctx and P <= n_all:      # padded lists prepared on the trace stream: views, no launch
            idx_pad, dst_pad = ctx['hit_idx_src'][:P], ctx['hit_idx_all'][:P]
        else:
            idx_pad = torch.cat([idx, idx[:1].expand(pad)]) if pad else idx
            dst_pad = torch.cat([idx, idx.new_full((pad,), n_all)]) if pad else idx
        ctx = {k: v for k, v in ctx.items() if k not in ('hit_idx_all', 'hit_idx_src', 'hit_count_host')}
        mat = self.model.envmap_material_network
        # everything a capture bakes in: shapes, the python-side switches of the material network, the loss's alpha
        rn = self.model.rendering_network
        # (a radiance network that runs without autograd is packed at capture time only: a capture is good for the parameter
        # versions it saw - load_state_dict in the middle of a run gets a new one)
        rver = sum(q._version for q in ops.param_list(rn)) if getattr(rn, 'outputs_detached', False) else None
        key = (P, n_all, bool(getattr(mat, 'fake_roughness', False)), bool(getattr(mat, 'fake_specular', False)),
               float(self.loss.alpha), rver)
        g = self._graphs.get(key)
        if g is None:
            g = self._graphs[key] = _StepGraph(self, ctx, idx_pad, dst_pad, ground_truth)
        else:
            g.load(ctx, idx_pad, dst_pad, ground_truth)
        g.graph.replay()
        for p, grad in g.grads:       # an eager step in between may have re-pointed .grad
            p.grad = grad
        if not g.updates_in_graph:
            self._update(g.lo['loss'])
        return g.out, g.lo

    def __call__(self, model_input, ground_truth, next_input=None):
        """next_input (optional): the model_input of the following call, or a list of the following calls' inputs in
        order (the same dict objects must then be passed to them) - their rays are traced concurrently with this step's
        tail, see prefetch_trace."""
        self._pre_iteration()
        ctx = None
        self._audit_seen = len(getattr(getattr(self.model, 'implicit_network', None), 'coarse_audit_events', ()))
        if next_input is not None:
            m = self.model
            upcoming = list(next_input) if isinstance(next_input, (list, tuple)) else [next_input]
            queued = [e[0] for e in self._prefetch]
            mine = bool(queued) and queued[0] is model_input
            expected = ([model_input] if mine else []) + upcoming
            if len(queued) > len(expected) or any(a is not b for a, b in zip(queued, expected)):
                self._prefetch, queued, mine = [], [], False        # the caller changed its mind about the coming batches
                expected = upcoming
            if not mine and m.training and getattr(m, 'state_freeze_geo', False):
                with self._min_sdf_schedule(self.cur_iter):
                    ctx = m.trace_head(model_input)         # own trace first: the tracer's random draws keep their order
            # enqueued BEFORE this batch's own (earlier enqueued) trace is waited for: the trace streams then always have
            # the next trace(s) queued behind / beside the running one and never idle while the host checks and launches
            todo = expected[len(queued):]
            it_next = self.cur_iter + (0 if mine else 1) + len(queued)       # the iteration todo[0] belongs to
            G = self.trace_group_for(model_input)
            if G <= 1:
                for inp in todo:
                    with self._min_sdf_schedule(it_next):
                        self.prefetch_trace(inp)
                    it_next += 1
            else:
                while len(todo) >= G:
                    with self._min_sdf_schedule(*range(it_next, it_next + G)):
                        self.prefetch_group(todo[:G])
                    todo = todo[G:]
                    it_next += G
                # never let the queue run dry: when no traced batch would be left for the next call, trace what there is
                left = len(self._prefetch) - (1 if self._prefetch and self._prefetch[0][0] is model_input else 0)
                if left == 0 and todo:
                    with self._min_sdf_schedule(*range(it_next, it_next + len(todo))):
                        self.prefetch_group(todo)
        if ctx is None:
            ctx = self._take_prefetched(model_input)
        if self.graph and self._eager_steps >= self.graph_after and self.model.training:
            res = self._graph_step(model_input, ground_truth, ctx)
            if res is not None:
                if self.secondary_train_interval > 0 and self.cur_iter % self.secondary_train_interval == 0:
                    self.train_with_secondary(res[0])
                self._note_sync_audit()
                self._post_iteration()
                return res
        self._eager_steps += 1
        if ctx is not None:
            out = self.model.shade_tail(ctx, _hit_index(ctx))
        else:
            with self._min_sdf_schedule(self.cur_iter):
                out = self.model(model_input)
        lo = self.loss(out, ground_truth)
        self._zero_grads()
        if lo['loss'].requires_grad:        # a slice without a single hit (and no background term) has nothing to
            lo['loss'].backward()           # differentiate; the reference wraps backward in try/except (:764-770)
        self._update(lo['loss'])
        if self.secondary_train_interval > 0 and self.cur_iter % self.secondary_train_interval == 0:
            self.train_with_secondary(out)
        self._note_sync_audit()
        self._post_iteration()
        return _detached(out), _detached(lo)

    def _note_sync_audit(self):
        """Audit events raised by this step's SYNCHRONOUS traces (its own primary trace when nothing was enqueued ahead, the
        secondary trace inside the tail): RayTracing.forward has already repeated those traces without the failed bound; what
        is left to do here is to record the event for the runner's log and to drop the traces enqueued AHEAD under the bound
        that no longer holds (they are traced again when their batch comes up)."""
        net = getattr(self.model, 'implicit_network', None)
        if net is None or not hasattr(net, 'coarse_audit_events'):
            return
        fresh = net.coarse_audit_events[self._audit_seen:]
        self._audit_seen = len(net.coarse_audit_events)
        known = {(k, o, b) for _, k, o, b in self.coarse_events[-len(fresh) - 8:]} if fresh else set()
        for kind, observed, bound in fresh:
            if (kind, observed, bound) not in known:
                self.coarse_events.append((self.cur_iter, kind, observed, bound))
        if any(kind in ('disabled', 'lipschitz_disabled') for kind, _, _ in fresh) and self._prefetch:
            self._prefetch = []
            self.retraced_steps += 1

    def train_with_secondary(self, model_outputs):
        """L1(sg_rgb, idr_rgb) at secondary hit points, seen from the direction they were hit from
        (idr_train.py:804-852): ties the material/light decomposition to the radiance field where the camera
        never looks."""
        pts, mask, dirs = (model_outputs.get(k) for k in ('secondary_points', 'secondary_mask', 'secondary_dir'))
        dev = self.nonfinite_steps.device
        # Whether this step's collective runs is decided by CONFIGURATION (the render type produces secondary rays, several
        # ranks train), never by data: a rank whose pixel slice has no primary hit gets no secondary outputs at all
        # (shade_tail: ret = {}), one with hits may have no secondary hit - both still enter the all-reduce with zero
        # gradients and step their optimizers like their peers.  Returning here would pair this rank's NEXT all-reduce
        # with its peers' current one (fixed-size buffers: silently), averaging gradients of different steps.
        has_secondary = getattr(self.model, 'render_type', 'sg') != 'sg'
        none = pts is None or mask is None or dirs is None
        if not has_secondary or (none and self.world_size <= 1):
            return None
        idx = torch.zeros(0, dtype=torch.long, device=dev) if none else \
            torch.nonzero(mask.reshape(-1)).flatten()[:self.secondary_batch_size]
        if idx.numel() == 0 and self.world_size <= 1:
            return None
        self._zero_grads()
        loss = None
        if idx.numel() > 0:
            p = pts.detach().reshape(-1, 3).index_select(0, idx)
            d = dirs.detach().reshape(-1, 3).index_select(0, idx)
            n = p.shape[0]
            ret = self.model({'points': p.unsqueeze(1).expand(n, self.num_rays, 3),
                              'ray_dirs': d.unsqueeze(1).expand(n, self.num_rays, 3)}, with_point=True)
            loss = torch.nn.functional.l1_loss(ret['sg_rgb_values'], ret['idr_rgb_values'])
            loss.backward()
        # several ranks: whether this rank has secondary hits is data, whether the collective runs must not be - a rank
        # without any still enters the all-reduce (with zero gradients) and steps its optimizers like its peers
        self._update(loss if loss is not None else torch.zeros((), device=dev))
        return loss
