"""Does the zero-padded stand-in geometry of configs 3-5 flatter the evaluators?  Its 512-wide SDF net holds 98 % zero weights;
a power-limited part multiplies zeros cheaply.  (1) tile time of both evaluators on random points for three nets: the
geometric-init sphere (dense), the stand-in zero-padded ('bowl'), the stand-in replicated ('bowl_dense': same function, no zero
weight).  (2) config 3's step with either embedding, board power beside it."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from nefii_amd import _lib, ops, synthetic as syn
from oracle import nets

dev = 'cuda'
mc = syn.model_conf('conf')
specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
g = torch.Generator().manual_seed(1)
n = 12 * 256 * 64
x = (torch.randn(n, 3, generator=g) * 0.4).cuda()
for tag, sd in (('sphere (geometric init, dense)', syn.make_state_dict(mc, seed=0, bumpy=0.004)),
                ('bowl, zero-padded', syn.make_state_dict(mc, seed=0, scene='bowl')),
                ('bowl, replicated (dense)', syn.make_state_dict(mc, seed=0, scene='bowl_dense'))):
    pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, dev, f16x3=True)
    ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
    pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
    res = []
    for coarse in (False, True):
        for _ in range(3):
            ops.sdf_eval(pm, x, coarse=coarse)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.sdf_eval(pm, x, coarse=coarse)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3 / 12)
    print('%-34s split %.1f us per 64-query tile, single pass %.1f us' % (tag, res[0], res[1]), flush=True)

lib = _lib.lib()
args = argparse.Namespace(repeats=3, scaling='weak')
for scene in ('bowl', 'bowl_dense', 'bowl', 'bowl_dense'):
    for w in ('cfg3',):
        syn.WORKLOADS[w]['scene'] = scene
        r = bench.run_workload(w, args, 10, 3, 0, 1, torch.device('cuda', 0), 'nccl', lib, side=False, power=True)
        print('%s %-11s %.1f ms per step %s  frac_kernel %.3f  board power %s  loss %.4f' % (
            w, scene, r['ms_per_step'], ['%.1f' % v for v in r['ms_per_step_repeats']], r['roofline']['frac'],
            {k: round(v) for k, v in (r['roofline']['board_power'] or {}).items() if k.endswith('_w')}, r['config']['loss']), flush=True)
