"""Where does the HOST spend a training step (cProfile over N steps after a warm-up)?  Blocking calls (torch.nonzero, .item(), .cpu(),
Event.synchronize, pageable copies) show up with the time the host WAITED in them.

    python tools/host_profile.py <workload> [steps=200] [warmup=30]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, '/root/repo')
import torch
from nefii_amd import conf, synthetic as syn
from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
from nefii_amd.training.step import TrainStep
name = sys.argv[1]
w = dict(syn.WORKLOADS[name]); mc = syn.model_conf(w['model'])
m = IDRNetwork(conf.from_dict(mc)); m.load_state_dict(syn.make_state_dict(mc, seed=0, scene=w.get('scene'))); m = m.to('cuda'); m.freeze_geometry(); m.train()
inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1)
inp = {k: v.cuda() for k, v in inp.items()}; gt = {'rgb': gt.cuda()}
indirect = mc.get('render_type', 'sg') != 'sg'
st = TrainStep(m, syn.loss_conf(w['model']), graph=not indirect, secondary_train_interval=10 if indirect else 0, secondary_batch_size=1024, num_rays=w['num_rays'])
nxt = [inp] * st.preferred_lookahead(inp)
WARM = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for _ in range(WARM): st(inp, gt, nxt)
torch.cuda.synchronize()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t0 = time.perf_counter()
for _ in range(N): st(inp, gt, nxt)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('%s: host loop %.3f ms/step, with final sync %.3f ms/step' % (name, t_host / N * 1e3, t_all / N * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(N): st(inp, gt, nxt)
pr.disable(); torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats('cumulative'); ps.print_stats(45)
