"""A long training run of one BASELINE workload under the DEFAULT schedule (traces of the coming batches beside the tail,
secondary-consistency step every 10 iterations for the MC workloads), several distinct batches cycled: does every step
survive the NaN guard, does the loss go down?  (VERDICT r2 missing #1: round 2's config 3 lost a third of its steps.)

    python tools/long_train.py [workload=cfg3] [steps=300] [batches=4] [out.json]

Prints / writes: steps cancelled by the guard, the loss curve in windows of 25 steps, ms per step."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from nefii_amd import conf, synthetic as syn
from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
from nefii_amd.training.step import TrainStep


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    out_path = sys.argv[4] if len(sys.argv) > 4 else None
    dev = 'cuda:0'
    w = dict(syn.WORKLOADS[name])
    mc = syn.model_conf(w['model'])
    torch.manual_seed(7)
    m = IDRNetwork(conf.from_dict(mc))
    m.load_state_dict(syn.make_state_dict(mc, seed=0, scene=w.get('scene')), strict=True)
    m = m.to(dev)
    m.freeze_geometry()
    m.ray_tracer.trace_tier = os.environ.get('NEFII_TRACE_TIER', '1') != '0'      # the per-run switch, as bench.py sets it
    m.train()
    batches = []
    for b in range(nb):
        inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=100 + b)
        # a learnable target: a smooth function of the pixel
        uv = inp['uv'][0].reshape(inp['uv'].shape[1], -1, 2).mean(1) / 800.0 if inp['uv'].dim() == 4 else inp['uv'][0] / 800.0
        gt = torch.stack([0.25 + 0.5 * uv[:, 0], 0.3 + 0.4 * uv[:, 1], 0.5 + 0.3 * torch.sin(6.0 * uv[:, 0])], dim=-1)[None]
        batches.append(({k: v.to(dev) for k, v in inp.items()}, {'rgb': gt.to(dev)}))
    indirect = mc.get('render_type', 'sg') != 'sg'
    st = TrainStep(m, syn.loss_conf(w['model']), secondary_train_interval=10 if indirect else 0, secondary_batch_size=1024,
                   num_rays=w['num_rays'], graph=not indirect)
    look = st.preferred_lookahead(batches[0][0])
    losses = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        inp, gt = batches[it % nb]
        nxt = [batches[(it + 1 + j) % nb][0] for j in range(look)]
        out, lo = st(inp, gt, nxt)
        losses.append(lo['loss'].detach().clone())
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    losses = torch.stack(losses).float().cpu()
    win = 25
    curve = [losses[i:i + win].mean().item() for i in range(0, steps, win)]
    res = {'workload': name, 'steps': steps, 'distinct_batches': nb, 'trace_lookahead': look,
           'secondary_train_interval': 10 if indirect else 0,
           'steps_cancelled_by_the_nan_guard': int(st.nonfinite_steps.item()),
           'all_losses_finite': bool(torch.isfinite(losses).all()),
           'loss_mean_per_%d_steps' % win: curve, 'ms_per_step': sec / steps * 1e3}
    print(json.dumps(res))
    if out_path:
        with open(out_path, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
