"""Per-round anatomy of the tracer on a workload: queries per round and the eval launch durations
(HIP events on the launch stream).  Usage: python tools/trace_rounds.py [cfg2] [train|eval]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import _lib, conf, synthetic as syn
from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork

wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
mode = sys.argv[2] if len(sys.argv) > 2 else 'train'
w = dict(syn.WORKLOADS[wl])
mc = syn.model_conf(w['model'])
sd = syn.make_state_dict(mc, seed=0, scene=w.get('scene'))
dev = torch.device('cuda:0')
m = IDRNetwork(conf.from_dict(mc))
m.load_state_dict(sd)
m = m.to(dev)
m.freeze_geometry()
m.ray_tracer.trace_tier = os.environ.get('NEFII_TRACE_TIER', '1') != '0'      # the per-run switch, as bench.py sets it
m.train(mode == 'train')
inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1)
inp = {k: v.to(dev) for k, v in inp.items()}
lib = _lib.lib()
m.ray_tracer.collect_counters = True
m.ray_tracer.adaptive_rounds = False
from nefii_amd.utils import rend_util
uv = inp['uv'] if inp['uv'].dim() == 3 else inp['uv'].reshape(1, -1, 2)
dirs, cam = rend_util.get_camera_params(uv, inp['pose'], inp['intrinsics'])
om = torch.ones(dirs.shape[1], dtype=torch.bool, device=dev)
for it in range(3):
    lib.nefii_trace_profile_enable(1)
    with torch.no_grad():
        m.ray_tracer(m.implicit_network, cam, om, dirs)
    torch.cuda.synchronize()
    buf = (ctypes.c_float * 256)()
    n = lib.nefii_trace_profile_launches(buf, 256)
    lib.nefii_trace_profile_enable(0)
cnt = m.ray_tracer.last_counters.cpu().tolist()
tot = 0.0
print('round  singles  dense  tri(consumed)  refined  coarse quarter rows  tier queries(repeated)  staged 2nd stage | split-precision queries  single-pass queries |   ms')
for r in range(n):
    c = cnt[r]
    split = c[0] + c[1] * 100 + c[7] + c[4]
    coarse = c[5] * 25 + c[9] + c[11]
    tot += buf[r]
    if split or coarse:
        print('%4d %8d %6d %5d(%6d) %8d %11d %13d(%6d) %16d | %23d %19d | %6.3f' % (r, c[0], c[1], c[2], c[3], c[4], c[5], c[9], c[10],
                                                                                   c[11], split, coarse, buf[r]))
print('total eval ms %.3f over %d launches (a launch = the round\'s split-precision dispatches + its coarse dispatch)' % (tot, n))
