"""The register-resident coarse evaluator ("16c", csrc/sdf_tile_c.h) against the split-precision evaluator, the fp64 oracle and
- in a child process with NEFII_COARSE_X=0 - the "16s" evaluator it replaces: error envelope and time per 64 queries.
Usage: python tools/coarse_x_check.py [n_points]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import ops, synthetic as syn
from oracle import nets


def run(n):
    out = {}
    for name, scene in (('physg', None), ('conf', 'bowl')):
        mc = syn.model_conf(name)
        sd = syn.make_state_dict(mc, seed=0, bumpy=0.004) if scene is None else syn.make_state_dict(mc, seed=0, scene=scene)
        specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
        pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
        ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
        pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(n, 3, generator=g) * 0.4).cuda()
        split = ops.sdf_eval(pm, x, coarse=False)
        coarse = ops.sdf_eval(pm, x, coarse=True)
        torch.cuda.synchronize()
        ref = nets.sdf_forward({k: v.double() for k, v in sd.items()}, mc['implicit_network'], x[:4000].cpu().double())[:, 0]
        d = (coarse - split).abs()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ops.sdf_eval(pm, x, coarse=True)
        e0.record()
        for _ in range(10):
            ops.sdf_eval(pm, x, coarse=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out[name] = dict(max=d.max().item(), rms=d.pow(2).mean().sqrt().item(), finite=bool(torch.isfinite(coarse).all()),
                         vs64=(coarse[:4000].cpu().double() - ref).abs().max().item(),
                         split64=(split[:4000].cpu().double() - ref).abs().max().item(), ms=ms,
                         us_per_64=ms * 1e3 / (n / 64 / 256))
    return out


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12 * 256 * 64
    if os.environ.get('COARSE_X_CHILD'):
        print(repr(run(n)))
        sys.exit(0)
    res = {'16c': run(n)}
    env = dict(os.environ, NEFII_COARSE_X='0', COARSE_X_CHILD='1')
    res['16s'] = eval(subprocess.run([sys.executable, __file__, str(n)], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    for k, v in res.items():
        for name, r in v.items():
            print('%s %-6s n %d: max|coarse-split| %.3e rms %.3e  |coarse-fp64| %.3e  |split-fp64| %.3e  finite %s   %.3f ms = %.1f us per 64 queries per CU'
                  % (k, name, n, r['max'], r['rms'], r['vs64'], r['split64'], r['finite'], r['ms'], r['us_per_64']))
