"""Is a small config bound by its TRACES or by the step's tail?  (VERDICT r4 next #2 asked for device-side round control on the
premise that configs 1 / 2 are launch latency.)  The traces of the timed step, and nothing else: the same prefetch calls
TrainStep makes (grouped three batches per call on four streams for config 1, one batch per call on three streams for config
2, deferred round-prefix checks, no host sync inside), enqueued back to back with K calls in flight - ms per BATCH next to
bench.py's ms per step of the same workload.  If the two agree, the tail (and its launch count) is not what bounds the step.

    python tools/experiments/trace_only.py [cfg1|cfg2] [batches=600]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from nefii_amd import conf, synthetic as syn
from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
from nefii_amd.training.step import TrainStep

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg1'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 600
w = dict(syn.WORKLOADS[name])
mc = syn.model_conf(w['model'])
m = IDRNetwork(conf.from_dict(mc))
m.load_state_dict(syn.make_state_dict(mc, seed=0, scene=w.get('scene')))
m = m.to('cuda')
m.freeze_geometry()
m.train()
inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1)
inp = {k: v.cuda() for k, v in inp.items()}
st = TrainStep(m, syn.loss_conf(w['model']), graph=True)
G = st.trace_group_for(inp)
for in_flight in (1, 2, 3, 4, 6):
    def enqueue():
        if G > 1:
            st.prefetch_group([inp] * G)
        else:
            st.prefetch_trace(inp)
    for rep in range(2):            # the first pass settles the round guesses
        st._prefetch = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = 0
        while done < N:
            while len(st._prefetch) < in_flight * G:
                enqueue()
            # consume the oldest call's batches exactly as a step would: wait for its event, run its deferred checks
            for _ in range(G):
                st._take_prefetched(inp)
                done += 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / done * 1e3
    print('%s: traces only, %d call(s) of %d batch(es) in flight: %.3f ms per batch' % (name, in_flight, G, dt), flush=True)
