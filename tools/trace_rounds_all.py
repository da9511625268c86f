"""Per-round work of EVERY tracer call of one training forward (primary trace, secondary trace of the MC render types):
singles / dense / bisection / refined / coarse rays per round.  Usage: python tools/trace_rounds_all.py [cfg3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import conf, synthetic as syn
from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork

wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
w = dict(syn.WORKLOADS[wl])
mc = syn.model_conf(w['model'])
m = IDRNetwork(conf.from_dict(mc))
m.load_state_dict(syn.make_state_dict(mc, seed=0, scene=w.get('scene')))
m = m.to('cuda:0')
m.freeze_geometry()
m.ray_tracer.trace_tier = os.environ.get('NEFII_TRACE_TIER', '1') != '0'      # the per-run switch, as bench.py sets it
m.train()
inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1)
inp = {k: v.to('cuda:0') for k, v in inp.items()}
rt = m.ray_tracer
rt.collect_counters = True
rt.adaptive_rounds = False
calls = []
orig = rt.forward


def spy(*a, **k):
    r = orig(*a, **k)
    calls.append((a[3].shape if len(a) > 3 else k['ray_directions'].shape, rt.last_counters.cpu().clone()))
    return r


rt.forward = spy
with torch.no_grad():
    m(inp)
    calls.clear()
    m(inp)
torch.cuda.synchronize()
ns = 100
for shape, c in calls:
    c = c.long()
    tri_nodes = 7
    print('tracer call on rays %s: singles %d, dense rays %d, bisection evaluations %d, refined %d, coarse quarter rows %d (of them min-SDF: see '
          'rounds), dense searches entered %d' % (tuple(shape), c[:, 0].sum(), c[:, 1].sum(), c[:, 7].sum(), c[:, 4].sum(), c[:, 5].sum(), c[:, 6].sum()))
    print('   split-precision evaluations %d, single-pass samples %d' % (c[:, 0].sum() + c[:, 1].sum() * ns + c[:, 7].sum() + c[:, 4].sum(),
                                                                          c[:, 5].sum() * ((ns + 3) // 4)))
    for r in range(c.shape[0]):
        if c[r].sum() > 0:
            print('   round %2d: singles %7d dense %6d tri %6d refined %7d coarse quarter rows %6d entered %6d' % (
                r, c[r, 0], c[r, 1], c[r, 2], c[r, 4], c[r, 5], c[r, 6]))
