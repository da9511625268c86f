"""Parity protocol of the tiered sphere tracing (nefii_tracer_params.trace_tier; VERDICT r4 next #1a), part A: the tracer alone.

For each geometry (the trained stand-ins of configs 3 / 4, the thin-feature scene, round 4's replicated stand-in, the geometric-init
sphere, a bumpy one) and each ray
set (primary rays of the config's camera; secondary rays from the primary hits into a random hemisphere, as
pt_render_indirect_mlp starts them) the same rays are traced three ways -

    base   coarse pass, split-precision sphere tracing   (the default: every decision the split evaluator's)
    tier   the same + trace_tier = 1 for a list of (kappa, gate)
    oracle the CPU restatement of the reference, on the first ORACLE_RAYS rays

- and the table says what the tier changes: hit-mask flips, rays that enter / leave the bracket search, rays whose sphere
tracing ends at another iteration, |delta depth| of the rays that hit both ways without the argmin, and the same against the
oracle for BOTH (the base trace is itself ~sdf_threshold from the reference on some rays: the tier's figure is to be read
next to it); then what it buys: split-precision evaluations, single-pass evaluations, repeated queries, rounds.

    python tools/tier_parity.py [rays per set, default 32768]        (GPU; prints the table)

Part B is the GPU suite run with NEFII_TRACE_TIER=1 (the golden / shrunk-config / long-run tests against the oracle), part C
bench.py with and without it: tools/tier_round.sh runs all three.  Measurement infrastructure; imports the oracle as its
checker."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nefii_amd import ops, synthetic as syn
from oracle import nets, tracer

DEV = 'cuda:0'
ORACLE_RAYS = int(os.environ.get('TIER_ORACLE_RAYS', '3072'))
TIERS = [(2.0, 4.0), (2.0, 8.0), (3.0, 4.0), (2.0, 2.0)]


def build_sdf(mc, sd):
    specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
    pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, DEV, f16x3=True)
    ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
    pm.pack([w.to(DEV) for w in ws], [b.to(DEV) for b in bs])
    return pm


def camera_set(wl, n):
    w = syn.WORKLOADS[wl]
    rays = 64
    inp, _ = syn.make_inputs(max(4, n // rays // 4 * 4), w['image_hw'], w['focal'], w['cam_pos'], rays, seed=5)
    from oracle import renderer
    dirs, cam = renderer.camera_rays(inp['uv'].reshape(1, -1, 2), inp['pose'], inp['intrinsics'])
    dirs = dirs.reshape(-1, 3)
    return cam.reshape(1, 3).expand_as(dirs).contiguous(), dirs.contiguous()


def run(pm, mc, o, d, training, steps, tau, tier=None):
    kw = {}
    if tier is not None:
        kw = dict(trace_tier=1, tier_kappa=tier[0], tier_gate=tier[1])
    tp = ops.make_tracer_params(mc['ray_tracer'], training, 'f16x3w', bisect_levels=3, coarse_tau=tau, **kw)
    lin = torch.linspace(0, 1, steps=tp.n_steps).to(DEV)
    kept = []
    om = torch.ones(o.shape[0], dtype=torch.bool, device=DEV)
    pts, hit, dist, cnt = ops.trace_rays(pm, tp, o, d, om, lin, steps, want_counters=True, keep_workspace=kept)
    torch.cuda.synchronize()
    return dict(pts=pts, hit=hit, dist=dist, cnt=cnt.cpu().long(), it=ops.trace_iterations(kept))


def versus(a, b, argmin=None):
    """what differs between two traces of the same rays (dicts of run(), or the oracle's)"""
    flips = int((a['hit'] != b['hit']).sum())
    both = a['hit'] & b['hit']
    if argmin is not None:
        both = both & ~argmin
    dd = (a['dist'] - b['dist']).abs()[both]
    dp = (a['pts'] - b['pts']).abs().max(dim=1).values[both]
    if dd.numel() == 0:
        return flips, 0, 0.0, 0.0, 0.0, 0.0, 0.0
    return (flips, int(both.sum()), float(dd.max()), float(dd.mean()), float((dd > 1e-5).float().mean()),
            float((dd > 1e-4).float().mean()), float(dp.max()))


def work(c, ns=100):
    split, coarse = ops.executed_evals(c, ns)
    busy = torch.nonzero(c[:, [0, 1, 2, 4, 5, 9]].sum(dim=1)).flatten()
    return int(split.sum()), int(coarse.sum()), int(c[:, 9].sum()), int(c[:, 10].sum()), int(c[:, 0].sum()), \
        (int(busy[-1]) + 1 if busy.numel() else 0)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    torch.manual_seed(0)
    cases = [('cfg3 geometry (conf 8x512 TRAINED on the bowl scene: the headline stand-in)', 'conf', dict(scene='bowl_trained'), 'cfg3'),
             ('cfg4 geometry (neus 8x256 TRAINED on the bowl scene)', 'neus', dict(scene='bowl_trained'), 'cfg4'),
             ('thin-feature scene (conf 8x512 trained on the cube frame)', 'conf', dict(scene='frame_trained'), 'cfg3'),
             ('round 4\'s cfg3 geometry (conf 8x512, 8x64 fit replicated: bowl_dense)', 'conf', dict(scene='bowl_dense'), 'cfg3'),
             ('cfg2 geometry (physg 8x512, geometric-init sphere)', 'physg', dict(), 'cfg2'),
             ('bumpy 8x512 (physg, bumpy 0.004)', 'physg', dict(bumpy=0.004), 'cfg2')]
    for title, name, kw, wl in cases:
        mc = syn.model_conf(name)
        sd = syn.make_state_dict(mc, seed=0 if 'scene' in kw else 2, **kw)
        pm = build_sdf(mc, sd)
        tau = ops.calibrate_coarse_tau(pm)
        sdf = lambda x: nets.sdf_forward(sd, mc['implicit_network'], x)[:, 0]
        o, d = camera_set(wl, n)
        o, d = o.to(DEV), d.to(DEV)
        g = torch.Generator().manual_seed(9)
        steps = torch.rand(100, generator=g).to(DEV)
        base_eval = run(pm, mc, o, d, False, steps, tau)
        # secondary rays: from the hits, a random direction of the hemisphere about the normal
        hp = base_eval['pts'][base_eval['hit']][:n]
        _, _, nrm = ops.sdf_value_grad(pm, hp.contiguous())
        nrm = torch.nn.functional.normalize(nrm, dim=1)
        w = torch.nn.functional.normalize(torch.randn(hp.shape[0], 3, generator=g).to(DEV), dim=1)
        w = torch.where((w * nrm).sum(1, keepdim=True) < 0, -w, w).contiguous()
        print('=== %s: coarse_tau %.2e' % (title, tau))
        for set_name, oo, dd, training in (('primary, training mode', o, d, True), ('primary, eval mode', o, d, False),
                                           ('secondary (eval-mode recurrences, as the MC renderer traces them)', hp.contiguous(), w, False)):
            if oo.shape[0] == 0:
                continue
            base = run(pm, mc, oo, dd, training, steps, tau)
            again = run(pm, mc, oo, dd, training, steps, tau)
            assert torch.equal(base['dist'], again['dist']) and torch.equal(base['hit'], again['hit']), 'the base trace is not deterministic'
            m = min(ORACLE_RAYS, oo.shape[0])
            ref = tracer.trace(sdf, oo[:m].cpu(), dd[:m].cpu(), torch.ones(m, dtype=torch.bool), mc['ray_tracer'], training, steps.cpu())
            ref = dict(hit=ref['hit'], dist=ref['dists'], pts=ref['points'])
            argmin_ref = ~ref['hit'] if not training else ~ref['hit']
            cut = lambda r: dict(hit=r['hit'][:m].cpu(), dist=r['dist'][:m].cpu(), pts=r['pts'][:m].cpu())
            sb, cb, _, _, s0, rb = work(base['cnt'])
            print('--- %s: %d rays, hit fraction %.3f; base: %d split-precision + %d single-pass evaluations (%d of the split ones are '
                  'sphere-tracing queries), %d rounds' % (set_name, oo.shape[0], float(base['hit'].float().mean()), sb, cb, s0, rb))
            fo = versus(cut(base), ref, argmin_ref)
            print('    base vs oracle (%d rays): flips %d | both hit %d: |d depth| max %.2e mean %.2e, > 1e-5: %.4f, > 1e-4: %.4f, |d point| max %.2e'
                  % ((m,) + fo))
            for tier in TIERS:
                t = run(pm, mc, oo, dd, training, steps, tau, tier)
                st, ct, c9, c10, s0t, rt = work(t['cnt'])
                fb = versus(t, base)
                ft = versus(cut(t), ref, argmin_ref)
                its = int((t['it'] != base['it']).sum())
                print('    kappa %.2f gate %.0f: split %d (%.3f of base), single-pass %d; tier queries %d = %.3f of the sphere-tracing '
                      'queries, repeated %d (%.3f); rounds %d' % (tier[0], tier[1], st, st / max(sb, 1), ct, c9,
                                                                     c9 / max(c9 + s0t - c10, 1), c10, c10 / max(c9, 1), rt))
                print('        vs base  : flips %d, iteration changes %d | both hit %d: |d depth| max %.2e mean %.2e, > 1e-5: %.4f, > 1e-4: %.4f, '
                      '|d point| max %.2e' % ((fb[0], its) + fb[1:]))
                print('        vs oracle: flips %d | both hit %d: |d depth| max %.2e mean %.2e, > 1e-5: %.4f, > 1e-4: %.4f, |d point| max %.2e'
                      % ft)
                aud = float(t['cnt'][:, 8].contiguous().to(torch.int32).view(torch.float32).max())
                print('        audit: largest |single pass - split| among refined samples and repeated queries %.2e (bound %.2e)' % (aud, tau))


if __name__ == '__main__':
    main()
