"""IDRTrainRunner: the Step-2 training loop around the hot path, with the reference runner's constructor keywords,
experiment-directory layout and checkpoint format (code/training/idr_train.py:22-372, 612-802), so that a reference
checkpoint continues here and vice versa:

    <exps_folder>/<expname>/<timestamp>/checkpoints/{ModelParameters, IDROptimizerParameters, IDRSchedulerParameters,
        SGOptimizerParameters, SGSchedulerParameters}/{<epoch>, latest}.pth
    with {"epoch", "model_state_dict"} / {"epoch", "optimizer_state_dict"} / {"epoch", "scheduler_state_dict"}.

One process per GPU (RANK/LOCAL_RANK/WORLD_SIZE from the launcher; the reference's --local_rank also works); the
per-iteration work is training/step.py:TrainStep (forward, IDRLoss, backward, gradient all-reduce, 2x Adam, 2x
MultiStepLR, alpha milestones, roughness/specular warm-up, secondary-consistency step).  Not built: tensorboard
logging (vis_train writes its images as files instead, `plot_freq`), vis_test / plot_to_disk, camera optimisation, the view-diff pixel pairing - the reference's
image stack is not installed and none of it is on the hot path."""
import os
import sys
from datetime import datetime

import numpy as np
import torch
import torch.distributed as dist

from .. import conf as hocon
from ..utils import general as utils
from .step import TrainStep

SUBDIRS = {'model': 'ModelParameters', 'idr_opt': 'IDROptimizerParameters', 'idr_sched': 'IDRSchedulerParameters',
           'sg_opt': 'SGOptimizerParameters', 'sg_sched': 'SGSchedulerParameters'}


class IDRTrainRunner:
    def __init__(self, **kwargs):
        torch.set_default_dtype(torch.float32)
        self.local_rank = kwargs.get('local_rank', int(os.environ.get('LOCAL_RANK', -1)) if 'RANK' in os.environ else -1)
        self.multiprocessing = self.local_rank > -1
        if self.multiprocessing:
            torch.cuda.set_device(self.local_rank)
            if not dist.is_initialized():
                dist.init_process_group(backend=kwargs.get('dist_backend', 'nccl'))      # RCCL on ROCm
            self.device = torch.device('cuda', self.local_rank)
            self.world_size = dist.get_world_size()
        else:
            self.device = torch.device('cuda')
            self.world_size = 1
        self.rank = dist.get_rank() if self.multiprocessing else 0
        c = kwargs['conf']
        self.conf = c if isinstance(c, hocon.ConfigTree) else hocon.parse_file(c)
        self.batch_size = kwargs.get('batch_size', 1)
        self.nepochs = kwargs.get('nepochs', 2000)
        self.max_niters = kwargs.get('max_niters', 200001)
        self.exps_folder_name = kwargs.get('exps_folder_name', 'exps')
        self.expname = kwargs.get('expname', 'default')
        self.freeze_geometry = kwargs.get('freeze_geometry', False)
        self.freeze_idr = kwargs.get('freeze_idr', False)
        self.freeze_decompose_render = kwargs.get('freeze_decompose_render', False)
        self.freeze_light = kwargs.get('freeze_light', False)
        self.freeze_diffuse = kwargs.get('freeze_diffuse', False)
        self.ckpt_freq = kwargs.get('ckpt_freq', self.conf.get_int('train.ckpt_freq', default=5000))
        self.log_freq = kwargs.get('log_freq', 50)
        self.coordinate_type = kwargs.get('coordinate_type', 'mitsuba')         # idr_train.py:67 (the envmap image's axes)
        self.prefetch = kwargs.get('prefetch', True)       # TrainStep.prefetch_trace (frozen geometry only)
        if kwargs.get('train_cameras', False):
            raise NotImplementedError('camera optimisation is outside the Step-2 hot path')

        is_continue, timestamp = kwargs.get('is_continue', False), kwargs.get('timestamp', 'latest')
        self.expdir = os.path.join(self.exps_folder_name, self.expname)
        if is_continue and timestamp == 'latest':                                  # idr_train.py:76-92
            old = str(kwargs.get('old_expdir') or '') or self.expdir
            stamps = sorted(s for s in os.listdir(old) if '.' not in s) if os.path.exists(old) else []
            is_continue, timestamp = (True, stamps[-1]) if stamps else (False, None)
        if self.rank == 0:
            self.timestamp = kwargs.get('new_timestamp') or '{:%Y_%m_%d_%H_%M_%S}'.format(datetime.now())
            self.checkpoints_path = os.path.join(self.expdir, self.timestamp, 'checkpoints')
            for sub in SUBDIRS.values():
                os.makedirs(os.path.join(self.checkpoints_path, sub), exist_ok=True)
            if not isinstance(c, hocon.ConfigTree):
                with open(c) as f, open(os.path.join(self.expdir, self.timestamp, 'runconf.conf'), 'w') as g:
                    g.write(f.read())
            with open(os.path.join(self.expdir, self.timestamp, 'runcmd.txt'), 'w') as f:
                f.write('shell command : {0}'.format(' '.join(sys.argv)))

        ds_cls = kwargs.get('dataset_class') or self.conf.get_string('train.dataset_class')
        self.train_dataset = utils.get_class(ds_cls)(kwargs.get('gamma', 1.0), kwargs.get('data_split_dir', ''), False,
                                                     kwargs.get('subsample', 1), wo_mask=kwargs.get('wo_mask', False),
                                                     **kwargs.get('dataset_kwargs', {}))
        self.train_sampler_generator = torch.Generator()
        sampler = torch.utils.data.RandomSampler(self.train_dataset, generator=self.train_sampler_generator)
        self.train_dataloader = torch.utils.data.DataLoader(self.train_dataset, batch_size=self.batch_size,
                                                            collate_fn=self.train_dataset.collate_fn, sampler=sampler)
        self.n_batches = len(self.train_dataloader)
        # vis_train (idr_train.py:380-610): every `train.plot_freq` iterations rank 0 renders one training view in eval mode
        # and writes the frame's buffers (the reference sends the same images to tensorboard).  Off unless asked for:
        # `plot_freq` keyword (None: the conf's value when `plots=True`, else never).
        pf = kwargs.get('plot_freq')
        if pf is None and kwargs.get('plots', False):
            pf = self.conf.get_int('train.plot_freq', default=0)
        self.plot_freq = int(pf or 0)
        self.plot_num_rays = kwargs.get('plot_num_rays', 1)
        self.memory_capacity_level = kwargs.get('memory_capacity_level', 18)

        model_cls = kwargs.get('model_class') or self.conf.get_string('train.model_class')
        self.model = utils.get_class(model_cls)(conf=self.conf.get_config('model')).to(self.device)
        for key, part in (('pretrain_geometry_path', ('implicit_network',)),         # idr_train.py:205-244
                          ('pretrain_idr_rendering_path', ('rendering_network',)),
                          ('pretrain_diffuse_path', ('envmap_material_network', 'diffuse_albedo_layers'))):
            path = kwargs.get(key)
            if path and os.path.exists(path):
                sd = torch.load(path, map_location=self.device)['model_state_dict']
                sd = {k: v for k, v in sd.items() if tuple(k.split('.')[:len(part)]) == part}
                full = self.model.state_dict()
                full.update(sd)
                self.model.load_state_dict(full)
        if kwargs.get('light_sg_path') and os.path.exists(kwargs['light_sg_path']):
            self.model.envmap_material_network.load_light(kwargs['light_sg_path'])
        self.start_epoch = 0
        saved = {}
        if is_continue:
            old = os.path.join(str(kwargs.get('old_expdir') or '') or self.expdir, timestamp, 'checkpoints')
            ck = str(kwargs.get('checkpoint', 'latest')) + '.pth'
            saved = {k: torch.load(os.path.join(old, sub, ck), map_location=self.device) for k, sub in SUBDIRS.items()}
            self.model.load_state_dict(saved['model']['model_state_dict'])
            self.start_epoch = saved['model']['epoch']
        # --geometry / --geometry_neus override whatever was loaded before them, the continued run's checkpoint
        # included (idr_train.py:294-306 come after the is_continue block :251-292)
        if str(kwargs.get('geometry', '')).endswith('.pth'):                        # idr_train.py:294-301
            sd = torch.load(kwargs['geometry'], map_location=self.device)['model_state_dict']
            full = self.model.state_dict()
            full.update({k: v for k, v in sd.items() if 'implicit_network' in k})
            self.model.load_state_dict(full)
        if str(kwargs.get('geometry_neus', '')).endswith('.pth'):                   # idr_train.py:303-306
            self.model.implicit_network.load_state_dict(
                torch.load(kwargs['geometry_neus'], map_location=self.device)['sdf_network_fine'])


        self.num_pixels = self.conf.get_int('train.num_pixels')
        self.num_rays = self.conf.get_int('train.num_rays', default=-1)
        if self.freeze_idr:                                                         # idr_train.py:620-638
            self.model.freeze_idr()
        elif self.freeze_geometry:
            self.model.freeze_geometry()
        if self.freeze_decompose_render:
            self.model.freeze_decompose_render()
        if self.freeze_light:
            self.model.envmap_material_network.freeze_light()
        if self.freeze_diffuse:
            self.model.envmap_material_network.freeze_diffuse()
        self.model.train()
        # Tiered sphere tracing (model/ray_tracing.py): a property of the RUN - the `trace_tier` keyword
        # (exp_runner's --trace_tier), else the conf's train.trace_tier, else what the model block / NEFII_TRACE_TIER set (off
        # by default).  Every trace of the run then uses the same arithmetic whatever its batch size or world size; the choice
        # is printed with the first log line and saved with the model checkpoint ("trace_tier").
        tt = kwargs.get('trace_tier')
        if tt is None and self.conf.get('train.trace_tier', None) is not None:
            tt = self.conf.get_bool('train.trace_tier')
        rt = getattr(self.model, 'ray_tracer', None)
        if tt is not None and rt is not None and os.environ.get('NEFII_TRACE_TIER', '') == '':
            rt.trace_tier = bool(tt)
        self.trace_tier = bool(rt.tier_for()) if rt is not None and hasattr(rt, 'tier_for') else False
        # likewise the staged bracket search of eval-mode traces (the MC renderer's secondary rays): opt-in per run
        # (`bracket_staged_eval` keyword / --bracket_staged_eval, conf train.bracket_staged_eval, NEFII_BRACKET_STAGED_EVAL=1)
        be = kwargs.get('bracket_staged_eval')
        if be is None and self.conf.get('train.bracket_staged_eval', None) is not None:
            be = self.conf.get_bool('train.bracket_staged_eval')
        if be is not None and rt is not None and os.environ.get('NEFII_BRACKET_STAGED_EVAL', '') == '':
            rt.bracket_staged_eval = bool(be)

        t = self.conf.get_config('train')
        self.step = TrainStep(
            self.model, dict(self.conf.get_config('loss')), idr_lr=t.get_float('idr_learning_rate'),
            sg_lr=t.get_float('sg_learning_rate'), world_size=self.world_size,
            secondary_train_interval=kwargs.get('secondary_train_interval', 0),
            secondary_batch_size=kwargs.get('secondary_batch_size', 1024), num_rays=self.num_rays,
            graph=kwargs.get('graph', True),
            idr_sched_milestones=t.get_list('idr_sched_milestones', default=[]),
            idr_sched_factor=t.get_float('idr_sched_factor', default=0.0),
            sg_sched_milestones=t.get_list('sg_sched_milestones', default=[]),
            sg_sched_factor=t.get_float('sg_sched_factor', default=0.0),
            alpha_milestones=t.get_list('alpha_milestones', default=[]), alpha_factor=t.get_float('alpha_factor', default=0.0),
            roughness_warmup=kwargs.get('roughness_warmup', -1), specular_warmup=kwargs.get('specular_warmup', -1),
            start_iter=self.start_epoch * self.n_batches,
            # The reference runs the tracer's min-SDF search on EVERY iteration (ray_tracing.py:71-97) and so does this runner
            # by default.  Under frozen geometry the search only feeds the VALUE of mask_loss (no gradient); the opt-in
            # `min_sdf_every=E` (or NEFII_MIN_SDF_EVERY=E) runs it on the iterations with cur_iter % E == 0 only - e.g. E =
            # log_freq: the logged line is then the reference's, but the losses of the OTHER iterations (returned by the
            # step, and what the NaN guard sees, idr_train.py:754) carry a mask_loss computed without the search:
            # they are not the reference's values.  Same gradients, parameters and RNG stream either way (TrainStep).
            min_sdf_every=(kwargs.get('min_sdf_every') or int(os.environ.get('NEFII_MIN_SDF_EVERY', '1')))
            if self.freeze_geometry else 1)
        self.loss = self.step.loss
        if saved:
            self.step.idr_optimizer.load_state_dict(saved['idr_opt']['optimizer_state_dict'])
            self.step.idr_scheduler.load_state_dict(saved['idr_sched']['scheduler_state_dict'])
            self.step.sg_optimizer.load_state_dict(saved['sg_opt']['optimizer_state_dict'])
            self.step.sg_scheduler.load_state_dict(saved['sg_sched']['scheduler_state_dict'])
            self.step.retensor_lr()
        self.history = []
        self._coarse_events_logged = 0
        # the reference's tensorboardX log (idr_train.py:114-115): an event file in <exps>/<expname>/<timestamp>/, written by
        # this package's own writer (utils/tb_writer.py: tensorboard / tensorboardX are not installable here); rank 0 only
        self.writer = None
        if self.rank == 0 and kwargs.get('tensorboard', True):
            from ..utils.tb_writer import SummaryWriter
            self.writer = SummaryWriter(os.path.join(self.expdir, self.timestamp))

    # ---- idr_train.py:329-372
    def save_checkpoints(self, epoch):
        if self.rank != 0:
            return
        st = self.step
        # never overwrite latest.pth with poisoned weights (the reference checks the loss before backward / step, so its
        # emergency checkpoint is usable, idr_train.py:754-757; TrainStep._update cancels such a step, this is the belt)
        bad = [k for k, v in self.model.state_dict().items() if v.dtype.is_floating_point and not torch.isfinite(v).all()]
        if bad:
            raise FloatingPointError('refusing to checkpoint non-finite parameters: %s' % ', '.join(bad[:4]))
        payload = {'model': {'epoch': epoch, 'model_state_dict': self.model.state_dict(), 'trace_tier': self.trace_tier},
                   'idr_opt': {'epoch': epoch, 'optimizer_state_dict': st.portable_state_dict(st.idr_optimizer)},
                   'idr_sched': {'epoch': epoch, 'scheduler_state_dict': st.idr_scheduler.state_dict()},
                   'sg_opt': {'epoch': epoch, 'optimizer_state_dict': st.portable_state_dict(st.sg_optimizer)},
                   'sg_sched': {'epoch': epoch, 'scheduler_state_dict': st.sg_scheduler.state_dict()}}
        for key, sub in SUBDIRS.items():
            for name in (str(epoch), 'latest'):
                torch.save(payload[key], os.path.join(self.checkpoints_path, sub, name + '.pth'))

    def vis_train(self, it):
        """one training view through the eval-mode full-frame path, written to <timestamp>/plots as the render script
        writes its frames (training/render.py:write_frame); the sampling state of the dataset is put back afterwards"""
        from . import render as R
        ds = self.train_dataset
        keep = (ds.sampling_idx, ds.sampling_rays)
        ds.sampling_idx, ds.sampling_rays = None, None
        try:
            if self.plot_num_rays > 1:
                ds.change_sampling_rays(self.plot_num_rays)
            view = (it // self.plot_freq) % len(ds)
            idx, sample, gt = ds.collate_fn([ds[view]])
            model_input = {k: v.to(self.device) for k, v in sample.items()}
            self.model.eval()
            out = R.render_frame(self.model, model_input, ds.total_pixels, num_rays=max(self.plot_num_rays, 1),
                                 memory_capacity_level=self.memory_capacity_level)
            plots = os.path.join(self.expdir, self.timestamp, 'plots')
            buf = R.write_frame(self.model, out, gt['rgb'].to(self.device), model_input['pose'], ds.img_res, plots, it)
            if self.writer is not None:                  # idr_train.py:516-550: the training view as images
                self.writer.add_image('train/panel-%d' % view, buf['panel'].clamp(0., 1.).permute(2, 0, 1), it)
                env = R.compute_envmap(self.model.envmap_material_network.get_light(), 256, 512,
                                       upper_hemi=getattr(self.model.envmap_material_network, 'upper_hemi', False),
                                       coordinate_type=getattr(self, 'coordinate_type', 'mitsuba'))
                env = env.clamp(min=0.).pow(1. / 2.2)      # tonemap_img + clip_img (idr_train.py:390-391,548-550)
                self.writer.add_image('train/envmap', env.clamp(0., 1.).permute(2, 0, 1), it)
                self.writer.flush()
        finally:
            ds.sampling_idx, ds.sampling_rays = keep
            self.model.train()

    def _resample(self):
        r = self.loss.r_patch
        ds = self.train_dataset
        if r < 1:                                                                   # idr_train.py:641-645
            ds.change_sampling_idx(self.num_pixels)
            if self.multiprocessing:
                sub = ds.sampling_idx.shape[0] // self.world_size
                last = self.rank == self.world_size - 1
                ds.sampling_idx = ds.sampling_idx[self.rank * sub:] if last else \
                    ds.sampling_idx[self.rank * sub: (self.rank + 1) * sub]
        else:                                                                       # :646-656
            n_patch = self.num_pixels // (4 * r * r)
            ds.change_sampling_idx_patch(n_patch, r)
            if self.multiprocessing:
                ds.scatter_sampling_idx_patch(self.rank, self.world_size, n_patch, r)

    # ---- idr_train.py:612-802
    def run(self):
        mse2psnr = lambda x: -10. * np.log(x + 1e-8) / np.log(10.)
        for epoch in range(self.start_epoch, self.nepochs + 1):
            np.random.seed(epoch)           # every rank draws the same patch list before cutting its slice
            self._resample()
            self.train_dataset.change_sampling_rays(self.num_rays)
            if self.step.cur_iter > self.max_niters:
                self.save_checkpoints(epoch)
                return self.history
            self.train_sampler_generator.manual_seed(epoch)
            resample = getattr(self.loss, 'sample_each_iter', False)

            def batches():       # batches of lookahead (3; 15 or 31 when small batches are traced in groups): TrainStep traces
                                 # them beside the tail of the current one
                window = []
                for item in self.train_dataloader:
                    if resample:
                        self._resample()
                    window.append((item[0], {k: v.to(self.device) for k, v in item[1].items()},
                                   {'rgb': item[2]['rgb'].to(self.device)}))
                    if len(window) == 1 + self.step.preferred_lookahead(window[0][1]):
                        yield window[0], (None if resample else [w[1] for w in window[1:]])
                        window.pop(0)
                while window:
                    yield window[0], (None if resample or len(window) == 1 else [w[1] for w in window[1:]])
                    window.pop(0)

            for data_index, ((indices, model_input, ground_truth), next_input) in enumerate(batches()):
                it = self.step.cur_iter
                if self.rank == 0 and it % self.ckpt_freq == 0:                      # :695-696 (before the step)
                    self.save_checkpoints(epoch)
                if self.rank == 0 and self.plot_freq > 0 and it % self.plot_freq == 0:    # :698-700
                    self.vis_train(it)
                _, lo = self.step(model_input, ground_truth, next_input if self.prefetch else None)
                if it % self.log_freq == 0:
                    loss = lo['loss'].item()
                    # :752-757.  The per-step check runs on the device (TrainStep._update: a step whose loss is not
                    # finite on any rank has its gradients zeroed on every rank before Adam runs); its counter is
                    # read here, at the logging points, and is the same number on every rank - they all stop together
                    if self.step.nonfinite_steps.item() > 0 or not np.isfinite(loss):
                        self.save_checkpoints(epoch)
                        raise FloatingPointError('nan/inf in loss at or before iteration %d' % it)
                    rec = {'iter': it, 'epoch': epoch, 'loss': loss, 'sg_rgb_loss': lo['sg_rgb_loss'].item(),
                           'sg_psnr': float(mse2psnr(lo['sg_rgb_loss'].item())),
                           'idr_lr': float(self.step.idr_optimizer.param_groups[0]['lr']),
                           'sg_lr': float(self.step.sg_optimizer.param_groups[0]['lr']), 'trace_tier': self.trace_tier}
                    # what the online audit of the tracer's coarse bound did since the last line (TrainStep.coarse_events: a
                    # bound raised, or the coarse pass switched off and the step's batch traced again) goes into the record
                    if len(self.step.coarse_events) > self._coarse_events_logged:
                        rec['coarse_audit_events'] = [list(e) for e in self.step.coarse_events[self._coarse_events_logged:]]
                        self._coarse_events_logged = len(self.step.coarse_events)
                        if self.rank == 0:
                            print('{0} coarse-pass audit: {1}'.format(self.expname, rec['coarse_audit_events']))
                    self.history.append(rec)
                    if self.writer is not None:          # the scalars of the reference's log() (idr_train.py:881-895)
                        for k in ('idr_rgb_loss', 'sg_rgb_loss', 'mask_loss', 'normalsmooth_loss', 'background_rgb_loss'):
                            if k in lo:
                                self.writer.add_scalar(k, lo[k].item(), it)
                        self.writer.add_scalar('loss', loss, it)
                        self.writer.add_scalar('sg_psnr', rec['sg_psnr'], it)
                        if 'idr_rgb_loss' in lo:
                            self.writer.add_scalar('idr_psnr', float(mse2psnr(lo['idr_rgb_loss'].item())), it)
                        for k in ('alpha', 'mask_weight', 'idr_rgb_weight', 'sg_rgb_weight', 'normalsmooth_weight'):
                            if hasattr(self.loss, k):
                                self.writer.add_scalar(k, float(getattr(self.loss, k)), it)
                        self.writer.add_scalar('idr_lr', rec['idr_lr'], it)
                        self.writer.add_scalar('sg_lr', rec['sg_lr'], it)
                        self.writer.flush()
                    if self.rank == 0:
                        print('{0} [{1}] ({2}/{3}): loss = {4:.6f}, sg_rgb_loss = {5:.6f}, sg_psnr = {6:.3f}, idr_lr = {7}, '
                              'sg_lr = {8}'.format(self.expname, epoch, data_index, self.n_batches, loss,
                                                   rec['sg_rgb_loss'], rec['sg_psnr'], rec['idr_lr'], rec['sg_lr']))
                if self.step.cur_iter > self.max_niters:
                    break
        return self.history
