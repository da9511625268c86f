"""Per-tile cost of the tracer's SDF tile evaluator (nefii_sdf_eval): n points = tiles_per_cu x 256 CUs x 64 rows.
Usage: python tools/eval_microbench.py [tiles_per_cu ...]   (NEFII_LIB_PATH selects an A/B build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import ops, synthetic as syn
from oracle import nets

mc = syn.model_conf('physg')
sd = syn.make_state_dict(mc, seed=0, bumpy=0.004)
specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
g = torch.Generator().manual_seed(1)
for tpc in [int(a) for a in sys.argv[1:]] or [1, 2, 8]:
    n = tpc * 256 * 64
    x = (torch.randn(n, 3, generator=g) * 0.4).cuda()
    for _ in range(3):
        out = ops.sdf_eval(pm, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        out = ops.sdf_eval(pm, x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ref = nets.sdf_forward({k: v.double() for k, v in sd.items()}, mc['implicit_network'], x[:2000].cpu().double())[:, 0]
    err = (out[:2000].cpu().double() - ref).abs().max().item()
    print('tiles/CU %3d  n %8d  %.3f ms  %.1f us per tile-slot  %.1f TFLOP/s algorithmic   max|err| vs fp64 %.2e'
          % (tpc, n, ms, ms * 1e3 / tpc, n * 3.67104e6 / ms / 1e9, err))
