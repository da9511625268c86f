"""How much cheaper is ONE trace of G batches than G traces of one batch (config 2: 4096 rays per batch)?  Times the
tracer alone: G sequential traces on one stream, G concurrent traces on G streams, one trace of G x 4096 rays.
Usage: python tools/group_trace_probe.py [G]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import conf, synthetic as syn
from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
from nefii_amd.utils import rend_util

G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = dict(syn.WORKLOADS['cfg2'])
mc = syn.model_conf(w['model'])
sd = syn.make_state_dict(mc, seed=0, scene=w.get('scene'))
dev = torch.device('cuda:0')
m = IDRNetwork(conf.from_dict(mc))
m.load_state_dict(sd)
m = m.to(dev)
m.freeze_geometry()
m.ray_tracer.trace_tier = os.environ.get('NEFII_TRACE_TIER', '1') != '0'      # the per-run switch, as bench.py sets it
m.train(True)
rays = []
for g in range(G):
    inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1 + g)
    inp = {k: v.to(dev) for k, v in inp.items()}
    dirs, cam = rend_util.get_camera_params(inp['uv'], inp['pose'], inp['intrinsics'])
    rays.append((cam, dirs, inp['object_mask'].reshape(-1)))
rt = m.ray_tracer
rt.concurrent = True            # 3 speculative bisection levels, as the prefetched traces run


def one(g):
    return rt(sdf=m.implicit_network, cam_loc=rays[g][0], object_mask=rays[g][2], ray_directions=rays[g][1])


def grouped():
    return rt(sdf=m.implicit_network, cam_loc=torch.cat([r[0] for r in rays]), object_mask=torch.cat([r[2] for r in rays]),
              ray_directions=torch.cat([r[1] for r in rays]))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


streams = [torch.cuda.Stream() for _ in range(G)]


def concurrent():
    cur = torch.cuda.current_stream()
    ev = cur.record_event()
    for g in range(G):
        streams[g].wait_event(ev)
        with torch.cuda.stream(streams[g]):
            one(g)
        cur.wait_event(streams[g].record_event())


print('%d batches of %d rays: sequential %.3f ms, concurrent on %d streams %.3f ms, one trace of %d rays %.3f ms' % (
    G, rays[0][1].shape[1], timed(lambda: [one(g) for g in range(G)]), G, timed(concurrent), G * rays[0][1].shape[1],
    timed(grouped)))
