"""Step-1 geometry fit (reference code/training/geometry_train.py:27-389): the SDF network is regressed, with an L1 loss,
onto signed-distance samples of a mesh; its checkpoint is what Step 2 loads through --pretrain_geometry_path /
--geometry and then freezes.

Same constructor keywords, experiment layout and checkpoint format as the reference runner (three sub-directories -
there is no SG optimiser in this step; checkpoints are keyed by the batch index, geometry_train.py:352-353).  The
iteration is `geometry_model(points)[:, 0:1]` -> L1 -> backward -> Adam -> MultiStepLR, with the network evaluated by the
fused MLP kernels and its weight gradients by `nefii_mlp_wgrad` (model/ImplicitNetwork.forward, ops.FusedMLPFn); samples
come from datasets/sdf_dataset.py on the GPU.  Not built: the tensorboard images of `vis_train` (renders through the
half-fitted geometry) - plotting is off the path, see idr_train.py's header."""
import argparse
import os
import sys
from datetime import datetime

import torch

from .. import conf as hocon
from ..datasets.sdf_dataset import SDFDataset
from ..utils import general as utils

SUBDIRS = {'model': 'ModelParameters', 'idr_opt': 'IDROptimizerParameters', 'idr_sched': 'IDRSchedulerParameters'}


class GeometryTrainRunner:
    def __init__(self, **kwargs):
        torch.set_default_dtype(torch.float32)
        if not torch.cuda.is_available():
            raise RuntimeError('nefii_amd: the geometry fit runs on the HIP kernels and needs a GPU')
        self.device = torch.device('cuda')
        c = kwargs['conf']
        self.conf = c if isinstance(c, hocon.ConfigTree) else hocon.parse_file(c)
        self.batch_size = kwargs.get('batch_size', 16384)
        self.nepochs = kwargs.get('nepochs', 1)
        self.max_niters = kwargs.get('max_niters', 200001)
        self.exps_folder_name = kwargs.get('exps_folder_name', 'exps')
        self.expname = kwargs.get('expname', 'default')
        self.sample_num = kwargs.get('sample_num', 100)
        self.log_freq = kwargs.get('log_freq', 50)

        is_continue, timestamp = kwargs.get('is_continue', False), kwargs.get('timestamp', 'latest')
        self.expdir = os.path.join(self.exps_folder_name, self.expname)
        if is_continue and timestamp == 'latest':                                   # geometry_train.py:60-76
            old = str(kwargs.get('old_expdir') or '') or self.expdir
            stamps = sorted(s for s in os.listdir(old) if '.' not in s) if os.path.exists(old) else []
            is_continue, timestamp = (True, stamps[-1]) if stamps else (False, None)
        self.timestamp = kwargs.get('new_timestamp') or '{:%Y_%m_%d_%H_%M_%S}'.format(datetime.now())
        self.checkpoints_path = os.path.join(self.expdir, self.timestamp, 'checkpoints')
        for sub in SUBDIRS.values():
            os.makedirs(os.path.join(self.checkpoints_path, sub), exist_ok=True)
        if not isinstance(c, hocon.ConfigTree):
            with open(c) as f, open(os.path.join(self.expdir, self.timestamp, 'runconf.conf'), 'w') as g:
                g.write(f.read())
        with open(os.path.join(self.expdir, self.timestamp, 'runcmd.txt'), 'w') as f:
            f.write('shell command : {0}'.format(' '.join(sys.argv)))

        self.train_dataset = SDFDataset(kwargs.get('mesh_path', ''), self.sample_num, self.max_niters,
                                        kwargs.get('scale_to_unit', True), device=self.device, mesh=kwargs.get('mesh'))
        # every item is a fresh draw (the reference pins the index for the same reason, utils/sampler.py:29-52); samples
        # are produced on the GPU, so the loader runs in this process whatever --num_workers says
        self.train_dataloader = torch.utils.data.DataLoader(self.train_dataset,
                                                            batch_size=max(1, self.batch_size // self.sample_num),
                                                            shuffle=False, collate_fn=self.train_dataset.collate_fn)

        model_cls = kwargs.get('model_class') or self.conf.get_string('train.model_class')
        self.model = utils.get_class(model_cls)(conf=self.conf.get_config('model')).to(self.device)
        self.geometry_model = self.model.implicit_network
        self.loss = torch.nn.L1Loss()
        t = self.conf.get_config('train')
        self.idr_optimizer = torch.optim.Adam(list(self.model.implicit_network.parameters()) +
                                              list(self.model.rendering_network.parameters()),
                                              lr=t.get_float('idr_learning_rate'))
        self.idr_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.idr_optimizer,
                                                                  t.get_list('idr_sched_milestones', default=[]),
                                                                  gamma=t.get_float('idr_sched_factor', default=0.0))
        for key, part in (('pretrain_geometry_path', 'implicit_network'),           # :152-176
                          ('pretrain_idr_rendering_path', 'rendering_network')):
            path = kwargs.get(key)
            if path and os.path.exists(path):
                sd = torch.load(path, map_location=self.device)['model_state_dict']
                full = self.model.state_dict()
                full.update({k: v for k, v in sd.items() if k.split('.')[0] == part})
                self.model.load_state_dict(full)
        if kwargs.get('light_sg_path') and os.path.exists(kwargs['light_sg_path']):
            self.model.envmap_material_network.load_light(kwargs['light_sg_path'])

        self.start_epoch = 0
        self.cur_iter = 0
        if is_continue:                                                             # :184-206 (epoch stays 0 there too)
            old = os.path.join(str(kwargs.get('old_expdir') or '') or self.expdir, timestamp, 'checkpoints')
            ck = str(kwargs.get('checkpoint', 'latest')) + '.pth'
            saved = {k: torch.load(os.path.join(old, sub, ck), map_location=self.device) for k, sub in SUBDIRS.items()}
            self.model.load_state_dict(saved['model']['model_state_dict'])
            self.idr_optimizer.load_state_dict(saved['idr_opt']['optimizer_state_dict'])
            self.idr_scheduler.load_state_dict(saved['idr_sched']['scheduler_state_dict'])
        if str(kwargs.get('geometry', '')).endswith('.pth'):                        # :208-215
            sd = torch.load(kwargs['geometry'], map_location=self.device)['model_state_dict']
            full = self.model.state_dict()
            full.update({k: v for k, v in sd.items() if 'implicit_network' in k})
            self.model.load_state_dict(full)
        self.ckpt_freq = kwargs.get('ckpt_freq', t.get_int('ckpt_freq', default=5000))
        self.history = []

    def save_checkpoints(self, epoch):                                              # :226-248
        payload = {'model': {'epoch': epoch, 'model_state_dict': self.model.state_dict()},
                   'idr_opt': {'epoch': epoch, 'optimizer_state_dict': self.idr_optimizer.state_dict()},
                   'idr_sched': {'epoch': epoch, 'scheduler_state_dict': self.idr_scheduler.state_dict()}}
        for key, sub in SUBDIRS.items():
            for name in (str(epoch), 'latest'):
                torch.save(payload[key], os.path.join(self.checkpoints_path, sub, name + '.pth'))

    def train_iteration(self, points, gt_sdf_value):                                # :358-376
        points = points.reshape(-1, 3).to(self.device)
        gt_sdf_value = gt_sdf_value.reshape(-1, 1).to(self.device)
        self.geometry_model.train()
        predict_sdf_value = self.geometry_model(points)[:, 0:1]
        loss = self.loss(predict_sdf_value, gt_sdf_value)
        self.idr_optimizer.zero_grad()
        loss.backward()
        self.idr_optimizer.step()
        return loss

    def run(self):                                                                  # :342-389
        self.cur_iter = self.start_epoch * len(self.train_dataloader)
        for epoch in range(self.start_epoch, self.nepochs + 1):
            if self.cur_iter > self.max_niters:
                self.save_checkpoints(epoch)
                return self.history
            for data_index, (points, gt_sdf_value) in enumerate(self.train_dataloader):
                if self.cur_iter % self.ckpt_freq == 0:
                    self.save_checkpoints(data_index)
                loss = self.train_iteration(points, gt_sdf_value)
                if self.cur_iter % self.log_freq == 0:
                    value = loss.item()
                    if value != value:
                        print('[WARNING] detect nan in loss! please check!')
                        self.save_checkpoints(epoch)
                        return self.history
                    self.history.append((self.cur_iter, value))
                    print('{} {}/{}: loss = {},  idr_lr = {}'.format(self.expname, self.cur_iter, self.max_niters, value,
                                                                     self.idr_scheduler.get_last_lr()[0]))
                self.cur_iter += 1
                self.idr_scheduler.step()
                if self.cur_iter > self.max_niters:
                    break
        self.save_checkpoints(self.nepochs)         # the reference leaves the last partial interval unsaved
        return self.history


def add_argument(parser):                                                           # :392-407 + exp_runner.py:12-70
    parser.add_argument('--conf', type=str, required=True)
    parser.add_argument('--mesh_path', type=str, default='')
    parser.add_argument('--sample_num', type=int, default=100, help='sample num')
    parser.add_argument('--num_workers', type=int, default=0, help='accepted for compatibility (samples come from the GPU)')
    parser.add_argument('--not_scale_to_unit', default=False, action='store_true')
    parser.add_argument('--batch_size', type=int, default=16384)
    parser.add_argument('--nepoch', type=int, default=1)
    parser.add_argument('--max_niter', type=int, default=200001)
    parser.add_argument('--expname', type=str, default='default')
    parser.add_argument('--exps_folder_name', type=str, default='exps')
    parser.add_argument('--is_continue', default=False, action='store_true')
    parser.add_argument('--old_expdir', type=str, default='')
    parser.add_argument('--timestamp', default='latest', type=str)
    parser.add_argument('--checkpoint', default='latest', type=str)
    parser.add_argument('--geometry', type=str, default='')
    parser.add_argument('--pretrain_geometry_path', type=str, default='')
    parser.add_argument('--pretrain_idr_rendering_path', type=str, default='')
    parser.add_argument('--light_sg_path', type=str, default='')
    parser.add_argument('--model_class', type=str, default='nefii_amd.model.implicit_differentiable_renderer.IDRNetwork')
    return parser


def main(argv=None):
    opt, _ignored = add_argument(argparse.ArgumentParser()).parse_known_args(argv)    # the Step-2 flags of run_s1.sh pass
    runner = GeometryTrainRunner(conf=opt.conf, batch_size=opt.batch_size, nepochs=opt.nepoch, max_niters=opt.max_niter,
                                 expname=opt.expname, exps_folder_name=opt.exps_folder_name, is_continue=opt.is_continue,
                                 old_expdir=opt.old_expdir, timestamp=opt.timestamp, checkpoint=opt.checkpoint,
                                 geometry=opt.geometry, pretrain_geometry_path=opt.pretrain_geometry_path,
                                 pretrain_idr_rendering_path=opt.pretrain_idr_rendering_path,
                                 light_sg_path=opt.light_sg_path, mesh_path=opt.mesh_path, sample_num=opt.sample_num,
                                 scale_to_unit=not opt.not_scale_to_unit, model_class=opt.model_class)
    runner.run()


if __name__ == '__main__':
    main()
