"""Per-tile cost of the tracer's SDF tile evaluators - split precision (nefii_sdf_eval), single pass (nefii_sdf_eval_coarse)
and, where the net has it, the split evaluator with its correction products on block-scaled fp8 (nefii_sdf_eval_fp8corr,
round 6) interleaved in one process: n points = tiles_per_cu x 256 CUs x 64 rows.
Usage: python tools/eval_microbench.py [tiles_per_cu ...]   (NEFII_LIB_PATH selects an A/B build; MODEL=neus: 8x256 net; SCENE=bowl_trained|frame_trained|bowl_dense|bowl: that geometry's weights)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import ops, synthetic as syn
from oracle import nets

mc = syn.model_conf(os.environ.get('MODEL', 'physg'))
scene = os.environ.get('SCENE')          # e.g. bowl_trained / frame_trained / bowl_dense / bowl: the tile time follows the operands
sd = syn.make_state_dict(mc, seed=0, bumpy=0.0 if scene else 0.004, scene=scene)
print('SDF net: %s, %s' % (os.environ.get('MODEL', 'physg'), scene or 'geometric-init sphere, bumpy 0.004'))
specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
g = torch.Generator().manual_seed(1)
for tpc in [int(a) for a in sys.argv[1:]] or [1, 2, 8]:
    n = tpc * 256 * 64
    x = (torch.randn(n, 3, generator=g) * 0.4).cuda()
    flops = sum(2 * sp.k_in * sp.n_out for sp in specs)
    ref = nets.sdf_forward({k: v.double() for k, v in sd.items()}, mc['implicit_network'], x[:2000].cpu().double())[:, 0]
    best = {}
    kinds = [('split', dict()), ('single pass', dict(coarse=True))] + ([('split, fp8 corr.', dict(fp8=True))] if ops.fp8corr_supported(pm) else [])
    for rnd in range(3):                   # interleaved rounds in one process (compare A/B within it only)
        for kind, kw in kinds:
            for _ in range(2):
                out = ops.sdf_eval(pm, x, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                out = ops.sdf_eval(pm, x, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            d = (out[:2000].cpu().double() - ref).abs()
            near = ref.abs() < 0.02
            best[kind] = min(best.get(kind, (1e9, 0, 0))[0], ms), d.max().item(), (d[near].max().item() if near.any() else 0.0)
    for kind, _ in kinds:
        ms, err, err_near = best[kind]
        print('%-17s tiles/CU %3d  n %8d  %.3f ms  %.1f us per 64-query tile-slot  %.1f TFLOP/s algorithmic   max|err| vs fp64 %.2e '
              '(within 0.02 of the surface %.2e)' % (kind, tpc, n, ms, ms * 1e3 / tpc, n * flops / ms / 1e9, err, err_near))
