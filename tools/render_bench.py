"""Eval-mode full-frame rendering throughput (the H2 caller of SURVEY.md section 8a; BASELINE config 5 shape: conf.conf
model, num_rays sub-pixel rays per pixel, memory_capacity_level 18 -> 2^18 // num_rays pixels per chunk) on a crop of
the frame.   usage: render_bench.py [model=conf] [num_rays=256] [pixels=8192]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import conf, synthetic as syn
from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
from nefii_amd.training.render import render_frame

name = sys.argv[1] if len(sys.argv) > 1 else 'conf'
num_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pixels = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
mc = syn.model_conf(name)
sd = syn.make_state_dict(mc, seed=0, scene=os.environ.get('SCENE', 'bowl_dense'))   # the non-convex stand-in of configs 3-5
m = IDRNetwork(conf.from_dict(mc))
m.load_state_dict(sd)
m = m.cuda().eval()
m.freeze_geometry()
m.ray_tracer.trace_tier = os.environ.get('NEFII_TRACE_TIER', '1') != '0'      # the per-run switch, as bench.py sets it
inp, _ = syn.make_inputs(pixels, (800, 800), 1111.0, (0., 0., 2.4), num_rays, seed=3)
inp = {k: v.cuda() for k, v in inp.items()}
S = inp['uv'].shape[1]
for _ in range(2):
    out = render_frame(m, inp, S, num_rays=num_rays, memory_capacity_level=18)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    out = render_frame(m, inp, S, num_rays=num_rays, memory_capacity_level=18)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({'workload': 'eval render, %s model, %d pixels x %d rays, chunks of %d pixels' % (name, S, num_rays, (1 << 18) // num_rays),
                  'seconds': dt, 'primary_rays_per_s': S * num_rays / dt, 'pixels_per_s': S / dt,
                  'full_800x800_frame_s': 640000 / (S / dt), 'hit_fraction': out['network_object_mask'].float().mean().item()}))
