"""Novel-view rendering of a trained experiment (reference code/scripts/render.py:30-521): loads
`<exps>/<expname>/<timestamp>/checkpoints/ModelParameters/<checkpoint>.pth`, renders every view of the test split from
`--start_index` on with `--num_rays` jittered rays per pixel, in chunks of `2^memory_capacity_level // num_rays` pixels,
and writes the per-frame buffers and the light's lat-long map under a new `<timestamp>/plots`.

    python -m nefii_amd.scripts.render --conf confs_sg/conf.conf --data_split_dir_test <scene>/test --expname robot \
        --is_continue --timestamp latest --checkpoint latest --num_rays 256 --memory_capacity_level 18
    (+ torchrun --nproc-per-node N: the chunks of a frame are dealt round-robin to the ranks, rank 0 writes)

The frame loop is training/render.py (chunk / shard / gather / merge contract of the reference, one fixed-shape
`dist.gather` per frame instead of pickled object lists); the training-only flags of the reference's run scripts are
accepted and ignored."""
import argparse
import os
import sys
from datetime import datetime

import torch
import torch.distributed as dist

from .. import conf as hocon
from ..training import render as R
from ..utils import general as utils


class RenderRunner:
    def __init__(self, **kwargs):
        torch.set_default_dtype(torch.float32)
        self.local_rank = kwargs.get('local_rank', -1)
        self.multiprocessing = self.local_rank > -1
        if self.multiprocessing:
            torch.cuda.set_device(self.local_rank)
            if not dist.is_initialized():
                dist.init_process_group(backend=kwargs.get('dist_backend', 'nccl'))
            self.device = torch.device('cuda', self.local_rank)
            self.world_size, self.rank = dist.get_world_size(), dist.get_rank()
        else:
            self.device = torch.device('cuda')
            self.world_size, self.rank = 1, 0
        c = kwargs['conf']
        self.conf = c if isinstance(c, hocon.ConfigTree) else hocon.parse_file(c)
        self.memory_capacity_level = kwargs.get('memory_capacity_level', 18)
        self.start_index = kwargs.get('start_index', 0)
        self.num_rays = kwargs.get('num_rays', 256)
        self.coordinate_type = kwargs.get('coordinate_type', 'mitsuba')
        self.exps_folder_name = kwargs.get('exps_folder_name', 'exps')
        self.expname = kwargs.get('expname', 'default')
        self.expdir = os.path.join(self.exps_folder_name, self.expname)
        old = str(kwargs.get('old_expdir') or '') or self.expdir
        timestamp = kwargs.get('timestamp', 'latest')
        if timestamp == 'latest':                                                   # render.py:75-90
            stamps = sorted(s for s in os.listdir(old) if '.' not in s) if os.path.exists(old) else []
            if not stamps:
                raise FileNotFoundError('no experiment to render under ' + old)
            timestamp = stamps[-1]
        ckpt = os.path.join(old, timestamp, 'checkpoints', 'ModelParameters', str(kwargs.get('checkpoint', 'latest')) + '.pth')
        self.timestamp = kwargs.get('new_timestamp') or '{:%Y_%m_%d_%H_%M_%S}'.format(datetime.now())
        self.plots_dir = os.path.join(self.expdir, self.timestamp, 'plots')
        if self.rank == 0:
            os.makedirs(self.plots_dir, exist_ok=True)
            with open(os.path.join(self.expdir, self.timestamp, 'runcmd.txt'), 'w') as f:
                f.write('shell command : {0}'.format(' '.join(sys.argv)))

        ds_cls = kwargs.get('dataset_class') or self.conf.get_string('train.dataset_class')
        sub = kwargs.get('subsample', 1) * kwargs.get('vis_subsample', 1)
        self.test_dataset = utils.get_class(ds_cls)(kwargs.get('gamma', 1.0), kwargs.get('data_split_dir_test', ''), False,
                                                    sub, **kwargs.get('dataset_kwargs', {}))
        model_cls = kwargs.get('model_class') or self.conf.get_string('train.model_class')
        self.model = utils.get_class(model_cls)(conf=self.conf.get_config('model')).to(self.device)
        saved = torch.load(ckpt, map_location=self.device)
        self.model.load_state_dict(saved['model_state_dict'])
        if kwargs.get('light_sg_path') and os.path.exists(kwargs['light_sg_path']):
            self.model.envmap_material_network.load_light(kwargs['light_sg_path'])
        self.model.freeze_geometry()
        self.model.eval()
        # tiered sphere tracing: per run (--trace_tier / trace_tier=...), else what the checkpoint was trained with, else the
        # model block / NEFII_TRACE_TIER (off by default)
        tt = kwargs.get('trace_tier')
        if tt is None:
            tt = saved.get('trace_tier')
        rt = getattr(self.model, 'ray_tracer', None)
        if tt is not None and rt is not None and os.environ.get('NEFII_TRACE_TIER', '') == '':
            rt.trace_tier = bool(tt)
        self.trace_tier = bool(rt.tier_for()) if rt is not None and hasattr(rt, 'tier_for') else False
        if kwargs.get('bracket_staged_eval') is not None and rt is not None and os.environ.get('NEFII_BRACKET_STAGED_EVAL', '') == '':
            rt.bracket_staged_eval = bool(kwargs['bracket_staged_eval'])

    def run(self):                                                                  # render.py:262-442
        ds = self.test_dataset
        ds.change_sampling_idx(-1)
        ds.change_sampling_rays(self.num_rays)
        written = []
        for index in range(self.start_index, len(ds)):
            idx, sample, gt = ds.collate_fn([ds[index]])
            model_input = {k: v.to(self.device) for k, v in sample.items()}
            out = R.render_frame(self.model, model_input, ds.total_pixels, num_rays=max(self.num_rays, 1),
                                 memory_capacity_level=self.memory_capacity_level, rank=self.rank,
                                 world_size=self.world_size)
            if self.rank == 0:
                R.write_frame(self.model, out, gt['rgb'].to(self.device), model_input['pose'], ds.img_res, self.plots_dir,
                              int(idx[0]))
                written.append(int(idx[0]))
        if self.rank == 0:
            R.write_envmap(self.model, self.plots_dir, coordinate_type=self.coordinate_type)
        return written


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--conf', type=str, required=True)
    p.add_argument('--data_split_dir', type=str, default='')
    p.add_argument('--data_split_dir_test', type=str, default='')
    p.add_argument('--gamma', type=float, default=1.0)
    p.add_argument('--subsample', type=int, default=1)
    p.add_argument('--vis_subsample', type=int, default=1)
    p.add_argument('--expname', type=str, default='')
    p.add_argument('--exps_folder_name', '--exps_folder', dest='exps_folder', type=str, default='exps')
    p.add_argument('--old_expdir', type=str, default='')
    p.add_argument('--is_continue', default=False, action='store_true')
    p.add_argument('--timestamp', default='latest', type=str)
    p.add_argument('--checkpoint', default='latest', type=str)
    p.add_argument('--memory_capacity_level', type=int, default=18)
    p.add_argument('--coordinate_type', type=str, default='mitsuba')
    p.add_argument('--light_sg', type=str, default='')
    p.add_argument('--start_index', type=int, default=0, help='start index')
    p.add_argument('--num_rays', type=int, default=256, help='ray number')
    p.add_argument('--local_rank', type=int, default=-1)
    p.add_argument('--model_class', type=str, default='nefii_amd.model.implicit_differentiable_renderer.IDRNetwork')
    p.add_argument('--dataset_class', type=str, default='')
    p.add_argument('--trace_tier', default=None, action='store_true',
                   help='tiered sphere tracing for this render (DESIGN.md 4f; default: what the checkpoint records, else off)')
    p.add_argument('--bracket_staged_eval', default=None, action='store_true',
                   help='stage the bracket search behind the measured slope bound (DESIGN.md section 4; default off)')
    opt, _ignored = p.parse_known_args(argv)
    local_rank = opt.local_rank if opt.local_rank > -1 else (int(os.environ['LOCAL_RANK']) if 'RANK' in os.environ else -1)
    RenderRunner(trace_tier=opt.trace_tier, bracket_staged_eval=opt.bracket_staged_eval, conf=opt.conf, data_split_dir_test=opt.data_split_dir_test or opt.data_split_dir, gamma=opt.gamma,
                 subsample=opt.subsample, vis_subsample=opt.vis_subsample, expname=opt.expname or 'default',
                 exps_folder_name=opt.exps_folder, old_expdir=opt.old_expdir, timestamp=opt.timestamp,
                 checkpoint=opt.checkpoint, memory_capacity_level=opt.memory_capacity_level,
                 coordinate_type=opt.coordinate_type, light_sg_path=opt.light_sg, start_index=opt.start_index,
                 num_rays=opt.num_rays, local_rank=local_rank, model_class=opt.model_class,
                 dataset_class=opt.dataset_class or None).run()


if __name__ == '__main__':
    main()
