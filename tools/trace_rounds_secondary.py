"""Per-round anatomy of the SECONDARY trace of a Monte-Carlo workload (the rays pt_render_indirect_mlp sends from the hit
points): one training-mode forward of the model, then the counters and launch durations of its last tracer call.
Usage: python tools/trace_rounds_secondary.py [cfg3]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import _lib, conf, synthetic as syn
from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork

wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
w = dict(syn.WORKLOADS[wl])
mc = syn.model_conf(w['model'])
sd = syn.make_state_dict(mc, seed=0, scene=w.get('scene'))
dev = torch.device('cuda:0')
m = IDRNetwork(conf.from_dict(mc))
m.load_state_dict(sd)
m = m.to(dev)
m.freeze_geometry()
m.ray_tracer.trace_tier = os.environ.get('NEFII_TRACE_TIER', '1') != '0'      # the per-run switch, as bench.py sets it
m.train()
inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1)
inp = {k: v.to(dev) for k, v in inp.items()}
lib = _lib.lib()
m.ray_tracer.collect_counters = True
m.ray_tracer.adaptive_rounds = False
for it in range(3):
    lib.nefii_trace_profile_enable(1)       # (re-armed per forward: the buffer then holds primary + secondary launches in order)
    with torch.no_grad():
        out = m(inp)
    torch.cuda.synchronize()
    buf = (ctypes.c_float * 512)()
    n = lib.nefii_trace_profile_launches(buf, 512)
    lib.nefii_trace_profile_enable(0)
cnt = m.ray_tracer.last_counters.cpu().tolist()        # the last tracer call of the forward: the secondary rays
rounds = len(cnt)
ms = list(buf[n - rounds:n]) if n >= rounds else [float('nan')] * rounds
n_sec = int(out['secondary_mask'].numel()) if out.get('secondary_mask') is not None else -1
print('%s: secondary trace, %d rays, %d hits' % (wl, n_sec, int(out['secondary_mask'].sum()) if n_sec >= 0 else -1))
print('round  singles  dense  tri(consumed)  refined  coarse quarter rows  tier queries(repeated) | split-precision queries  single-pass queries |   ms')
tot = [0, 0, 0.0]
for r in range(rounds):
    c = cnt[r]
    split = c[0] + c[1] * 100 + c[7] + c[4]
    coarse = c[5] * 25 + c[9] + c[11]
    if split or coarse:
        tot[0] += split
        tot[1] += coarse
        tot[2] += ms[r]
        print('%4d %8d %6d %5d(%6d) %8d %11d %13d(%6d) | %23d %19d | %6.3f' % (r, c[0], c[1], c[2], c[3], c[4], c[5], c[9], c[10], split,
                                                                              coarse, ms[r]))
print('total: %d split-precision + %d single-pass evaluations, eval ms %.3f' % (tot[0], tot[1], tot[2]))
