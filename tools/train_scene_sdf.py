#!/usr/bin/env python3
"""Train the benchmark geometry at FULL WIDTH with the repo's own Step-1 runner.

The stand-in geometry of configs 3-5 so far is an 8 x 64 fit replicated across the 512 columns ('bowl_dense': full-size
compute, rank-64 structure).  Tile time, clock and power follow the operand statistics on this power-limited part, and
coarse_tau is a measured bound that depends on the weights - so this tool regresses the conf's OWN network (8 x 512, PE6,
skip at 4, Softplus(100), weight norm; conf.conf / physg.conf; 8 x 256 d_out 257 for conf_neus.conf) to an analytic scene
(tools/scenes.py) with training/geometry_train.py:GeometryTrainRunner - the reference's Step-1 loop (geometry_train.py:342-389:
L1 on signed-distance samples, Adam + MultiStepLR, the fused MLP kernels forward / backward / weight gradient) - fed by
analytic samples instead of mesh samples, and writes the weights where nefii_amd.synthetic finds them:

    python tools/train_scene_sdf.py <bowl|frame> <conf|neus> [iterations] [out dir]
        -> <out dir>/scene_<scene>_sdf<width>.npz : lin{l}.weight_v as fp16 (3.7 MB at 8 x 512), weight_g / bias fp32

weight_v in halves is lossless for the reparameterisation's purpose (w = g v / |v|: the row scale g / |v| is an fp32 number,
so the effective weights keep full fp32 mantissas - no all-zero lo fragments that would flatter the evaluators) and defines
the stand-in: the geometry IS what the stored weights say.  GPU only (the runner refuses a CPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import torch

import scenes
from nefii_amd import conf as hocon, synthetic as syn
from nefii_amd.training.geometry_train import GeometryTrainRunner


class AnalyticSDFDataset(torch.utils.data.Dataset):
    """Items of `sample_num` positions with their analytic signed distance, drawn afresh every time (the interface of
    datasets/sdf_dataset.py:SDFDataset).  Half of the samples near the surface (kept with probability exp(-|d| / 0.03) from
    uniform draws in the ball - the sphere tracer marches THROUGH the neighbourhood of the surface and needs the field there,
    not only the zero set), a tenth of those pushed onto the surface by a Newton step and jittered with the two sigmas
    mesh-to-sdf uses (0.0025, 0.00025); the other half uniform in the ball of radius 1.05 (the tracer starts on the unit sphere)."""

    def __init__(self, fn, sample_num, max_iter_num, device, seed=0):
        self.fn, self.sample_num, self.max_iter_num, self.device = fn, sample_num, max_iter_num, device
        self.gen = torch.Generator(device=device).manual_seed(seed)

    def _ball(self, n):
        p = torch.empty(3 * n, 3, device=self.device).uniform_(-1.05, 1.05, generator=self.gen)
        return p[p.norm(dim=-1) < 1.05][:n]

    def __getitem__(self, idx):
        n = self.sample_num
        near = []
        while sum(x.shape[0] for x in near) < n // 2:
            p = self._ball(8 * n)
            d = self.fn(p)
            near.append(p[torch.rand(p.shape[0], device=self.device, generator=self.gen) < torch.exp(-d.abs() / 0.03)])
        near = torch.cat(near)[:n // 2]
        k = near.shape[0] // 10
        with torch.enable_grad():
            q = near[:k].clone().requires_grad_(True)
            d = self.fn(q)
            g, = torch.autograd.grad(d.sum(), q)
        surf = (q - d.unsqueeze(-1) * g / (g * g).sum(-1, keepdim=True).clamp_min(1e-12)).detach()
        sig = torch.where(torch.arange(k, device=self.device) % 2 == 0, 0.0025, 0.00025).unsqueeze(-1)
        near = torch.cat([surf + sig * torch.randn(k, 3, device=self.device, generator=self.gen), near[k:]])
        pts = torch.cat([near, self._ball(n - near.shape[0])])
        return pts, self.fn(pts).reshape(-1, 1)

    def __len__(self):
        return self.max_iter_num

    def collate_fn(self, batch_list):
        return tuple(torch.cat(entry, 0) for entry in zip(*batch_list))


def main():
    scene, model = sys.argv[1], sys.argv[2]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    out_dir = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, 'nefii_amd', 'assets')
    fn = scenes.SCENES[scene]
    mc = syn.model_conf(model)
    width = mc['implicit_network']['dims'][0]
    cfg = hocon.from_dict({
        'train': {'model_class': 'nefii_amd.model.implicit_differentiable_renderer.IDRNetwork',
                  # sdf.conf's rate; its milestones (25 k apart for 200 k iterations) scaled to this run's length
                  'idr_learning_rate': 5e-4, 'idr_sched_factor': 0.5,
                  'idr_sched_milestones': [iters // 2, 5 * iters // 8, 3 * iters // 4, 7 * iters // 8, 15 * iters // 16],
                  'ckpt_freq': 10 ** 9},
        'model': mc})
    torch.manual_seed(0)
    tetra = (np.array([[1, 1, 1], [-1, -1, 1], [-1, 1, -1], [1, -1, -1]], dtype=np.float64) * 0.3,
             np.array([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]], dtype=np.int64))       # the runner's constructor wants a mesh
    os.makedirs(out_dir, exist_ok=True)
    runner = GeometryTrainRunner(conf=cfg, batch_size=16384, nepochs=1, max_niters=iters, sample_num=16384, log_freq=500,
                                 exps_folder_name=os.path.join('/tmp', 'nefii_scene_fit'), expname='%s_%s' % (scene, model),
                                 mesh=tetra, scale_to_unit=False)
    ds = AnalyticSDFDataset(fn, 16384, iters, runner.device)
    runner.train_dataset = ds
    runner.train_dataloader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, collate_fn=ds.collate_fn)
    import time
    t0 = time.time()
    hist = runner.run()
    torch.cuda.synchronize()
    print('trained %d iterations in %.1f s (%.2f ms per iteration of 16384 samples); L1 loss %.5f -> %.5f' % (
        runner.cur_iter, time.time() - t0, (time.time() - t0) / max(runner.cur_iter, 1) * 1e3, hist[0][1], hist[-1][1]), flush=True)
    net = runner.model.implicit_network
    net.eval()
    with torch.no_grad():
        held = AnalyticSDFDataset(fn, 200000, 1, runner.device, seed=99)
        x, t = held[0]
        y = net(x)[:, 0:1]
        err = (y - t).abs()
        nearm = t.abs() < 0.02
        print('held-out (%d samples): mean |err| %.5f, near-surface (|d| < 0.02) mean %.5f, max %.5f; sign agreement %.5f' % (
            x.shape[0], err.mean().item(), err[nearm].mean().item(), err.max().item(), ((y > 0) == (t > 0)).float().mean().item()))
    sd = runner.model.state_dict()
    out = {}
    nl = len([k for k in sd if k.startswith('implicit_network.') and k.endswith('.bias')])
    for l in range(nl):
        v = sd['implicit_network.lin%d.weight_v' % l].detach().float().cpu()
        out['lin%d.weight_v' % l] = v.half().numpy()
        out['lin%d.weight_g' % l] = sd['implicit_network.lin%d.weight_g' % l].detach().float().cpu().numpy()
        out['lin%d.bias' % l] = sd['implicit_network.lin%d.bias' % l].detach().float().cpu().numpy()
    path = os.path.join(out_dir, 'scene_%s_sdf%d.npz' % (scene, width))
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')
    # what the STORED weights (weight_v rounded to halves) make of the scene
    with torch.no_grad():
        for l in range(nl):
            sd['implicit_network.lin%d.weight_v' % l].copy_(torch.from_numpy(out['lin%d.weight_v' % l]).float())
        runner.model.load_state_dict(sd)
        y = runner.model.implicit_network(x)[:, 0:1]
        err = (y - t).abs()
        print('stored weights: mean |err| %.5f, near-surface mean %.5f, max %.5f; sign agreement %.5f' % (
            err.mean().item(), err[nearm].mean().item(), err.max().item(), ((y > 0) == (t > 0)).float().mean().item()))


if __name__ == '__main__':
    main()
