"""Hunt for non-finite values in the training step (config 3 under trace prefetch).

    python tools/nan_hunt.py kernel            streamed MLP kernels on many tiles per workgroup: finite? equal to the f32 kernels?
    python tools/nan_hunt.py step [workload] [steps] [lookahead]
                                               training steps with device-side finite flags on every op's outputs (no host
                                               sync inside the steps: the flags are read back at the end), first offenders printed

Environment switches of the library (NEFII_MLP_STREAM, NEFII_VG_STREAM, NEFII_WGRAD_TR, NEFII_TRACE_STREAMS,
NEFII_MLP_PRECISION ...) apply as usual - one process per combination."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nefii_amd import conf, ops, synthetic as syn

dev = 'cuda'


def packed(mc, sd, kind, half):
    from oracle import nets
    F = mc['feature_vector_size']
    if kind == 'radiance':
        specs, enc, head = ops.radiance_specs(mc['rendering_network'], F)
        pm = ops.PackedMLP(specs, ops.ACT_RELU, head, enc, F, dev, half=half)
        ws, bs = zip(*[nets.linear_params(sd, 'rendering_network.lin%d' % l) for l in range(len(specs))])
    else:
        mcfg = mc['envmap_material_network']
        specs, enc = ops.material_specs(mcfg, F, 4 if mcfg.get('roughness_mlp') else 3)
        pm = ops.PackedMLP(specs, ops.ACT_ELU, ops.HEAD_SIGMOID, enc, F, dev, half=half)
        lp = 'envmap_material_network.diffuse_albedo_layers'
        ws = [sd['%s.%d.weight' % (lp, 2 * l)] for l in range(len(specs))]
        bs = [sd['%s.%d.bias' % (lp, 2 * l)] for l in range(len(specs))]
    pm.pack([w.to(dev) for w in ws], [b.to(dev) for b in bs])
    return pm


def kernel_probe():
    g = torch.Generator().manual_seed(5)
    for name in ('conf', 'neus', 'physg'):
        mc = syn.model_conf(name)
        sd = syn.make_state_dict(mc, seed=1)
        F = mc['feature_vector_size']
        for kind in ('radiance', 'material'):
            pm16, pm32 = packed(mc, sd, kind, 'f16x3'), packed(mc, sd, kind, False)
            for n in (4096, 40000, 139264):
                x = (torch.randn(n, 3, generator=g) * 0.4).to(dev)
                v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
                nr = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
                feat = (torch.randn(n, F, generator=g) * 0.3).to(dev) if F else None
                a16 = (pm16, x, v, nr, feat) if kind == 'radiance' else (pm16, x, None, None, feat)
                a32 = (pm32,) + a16[1:]
                o16, _, s16 = ops.mlp_forward(*a16, want_stash=True)
                o32, _, s32 = ops.mlp_forward(*a32, want_stash=True)
                bad_rows = (~torch.isfinite(o16).all(dim=1)).nonzero().flatten()
                bad_stash = int((~torch.isfinite(s16)).sum())
                err = (o16 - o32).abs().max().item() if bad_rows.numel() == 0 else float('nan')
                off = ((o16 - o32).abs().amax(dim=1) > 1e-3).nonzero().flatten()
                if off.numel():
                    print('       %d rows differ by more than 1e-3: rows mod 64 %s, tiles (row // 64) %s ...' % (
                        off.numel(), sorted(set((off % 64).tolist())), sorted(set((off // 64).tolist()))[:12]))
                print('%-6s %-9s n=%6d stream=%s: non-finite output rows %d (first %s, rows mod 64: %s), non-finite stash '
                      'entries %d, max |f16x3 - f32| %.3g' % (
                          name, kind, n, pm16.mlp_stream, bad_rows.numel(), bad_rows[:6].tolist(),
                          sorted(set((bad_rows % 64).tolist()))[:16], bad_stash, err), flush=True)
                d_out = torch.randn_like(o16) * 1e-6
                gs = ops.mlp_grad_scale(d_out)
                dz16 = ops.mlp_backward(pm16, d_out, s16, gs)
                dz32 = ops.mlp_backward(pm32, d_out, s32)
                # (compare what both kernels define: hidden layers in full, the last layer's n_out columns)
                L = len(pm16.specs)
                a_ = torch.cat([dz16[:L - 1].reshape(-1), dz16[L - 1][:, :pm16.specs[-1].n_out].reshape(-1)])
                b_ = torch.cat([dz32[:L - 1].reshape(-1), dz32[L - 1][:, :pm16.specs[-1].n_out].reshape(-1)])
                print('       backward: non-finite dz entries %d, rel err vs f32 %.3g' % (
                    int((~torch.isfinite(a_)).sum()), ((a_ - b_).norm() / b_.norm()).item()), flush=True)


class Watch:
    """device-side finite flags, read back once at the end"""

    def __init__(self):
        self.rows, self.step = [], -1

    def see(self, label, *tensors):
        for i, t in enumerate(tensors):
            if torch.is_tensor(t) and t.is_floating_point() and t.numel() > 0:
                self.rows.append((self.step, '%s[%d]' % (label, i), torch.isfinite(t.detach()).all()))

    def wrap(self, owner, name, label=None, outs=True, ins=False):
        fn = getattr(owner, name)
        label = label or name
        w = self

        def wrapped(*a, **k):
            if ins:
                w.see(label + '.in', *[x for x in a if torch.is_tensor(x)])
            r = fn(*a, **k)
            if outs:
                w.see(label, *(r if isinstance(r, (tuple, list)) else (r,)))
            return r
        setattr(owner, name, wrapped)

    def report(self):
        torch.cuda.synchronize()
        flags = torch.stack([f for _, _, f in self.rows]).cpu().tolist()
        bad = [(s, l) for (s, l, _), ok in zip(self.rows, flags) if not ok]
        print('watch: %d flags, %d non-finite' % (len(flags), len(bad)))
        by_step = {}
        for s, l in bad:
            by_step.setdefault(s, []).append(l)
        for s in sorted(by_step):
            print('  step %d: first offenders (enqueue order): %s' % (s, by_step[s][:12]))
        return by_step


def step_probe(workload='cfg3', steps=30, lookahead=3):
    from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
    from nefii_amd.training.step import TrainStep
    w = dict(syn.WORKLOADS[workload])
    mc = syn.model_conf(w['model'])
    sd = syn.make_state_dict(mc, seed=0, scene=w.get('scene'))
    lc = syn.loss_conf(w['model'])
    torch.manual_seed(1234)
    model = IDRNetwork(conf.from_dict(mc))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    model.freeze_geometry()
    model.train()
    inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1)
    inp = {k: v.to(dev) for k, v in inp.items()}
    gt = {'rgb': gt.to(dev)}
    indirect = mc.get('render_type', 'sg') != 'sg'
    step = TrainStep(model, lc, secondary_train_interval=10 if indirect else 0, secondary_batch_size=1024,
                     num_rays=w['num_rays'], graph=False)
    W = Watch()
    if os.environ.get('HUNT_WATCH', '1') != '0':
        W.wrap(ops, 'mlp_forward', ins=True)
        W.wrap(ops, 'mlp_backward', ins=True)
        W.wrap(ops, 'mlp_grad_scale', ins=True)
        W.wrap(ops, 'sdf_value_grad', ins=True)
        W.wrap(ops, 'sdf_eval', ins=True)
        W.wrap(ops, 'trace_rays', ins=True)
        W.wrap(ops, 'mis_sample', ins=True)
        W.wrap(ops, 'encode_inputs')
        for cls in (ops.FusedMLPFn, ops.McShadeFn, ops.EnvRadianceFn, ops.IdrLossFn, ops.SGRenderFn):
            for meth in ('forward', 'backward'):
                fn = getattr(cls, meth)

                def make(fn, label):
                    def wrapped(ctx, *a):
                        if label.endswith('backward'):
                            W.see(label + '.in', *a)
                        r = fn(ctx, *a)
                        W.see(label, *(r if isinstance(r, (tuple, list)) else (r,)))
                        return r
                    return staticmethod(wrapped)
                setattr(cls, meth, make(fn, cls.__name__ + '.' + meth))
    # HUNT_SAVE=1: keep stream-ordered clones of the sampler's inputs and outputs of every call (no host sync) and look at
    # the calls that produced non-finite values afterwards: which entries, from which inputs, and what a re-run gives
    saved = []
    if os.environ.get('HUNT_SAVE', '0') == '1':
        raw_mis = ops.mis_sample

        def mis(lgt, rough, normal, view, uniforms):
            r = raw_mis(lgt, rough, normal, view, uniforms)
            saved.append((W.step, [t.detach().clone() for t in (lgt, rough, normal, view, uniforms)], [t.clone() for t in r]))
            return r
        ops.mis_sample = mis
    nxt = [inp] * lookahead if lookahead > 0 else None
    losses = []
    for i in range(steps):
        W.step = i
        out, lo = step(inp, gt, nxt)
        W.see('out.sg_rgb', out['sg_rgb_values'])
        W.see('out.idr_rgb', out['idr_rgb_values'])
        W.see('loss', lo['loss'])
        for n_, p in model.named_parameters():
            if p.grad is not None:
                W.see('grad.' + n_, p.grad)
        losses.append(lo['loss'].detach())
    torch.cuda.synchronize()
    print('%s: %d steps, lookahead %d, nonfinite_steps = %d' % (workload, steps, lookahead, int(step.nonfinite_steps.item())))
    print('losses:', ['%.5f' % float(x) for x in torch.stack(losses).cpu()])
    if W.rows:
        W.report()
    if saved:
        ops.mis_sample = raw_mis
        shown = 0
        for stp, ins, outs in saved:
            wi, own, tab = outs
            bad = ~torch.isfinite(tab).all(dim=-1)          # [3, n]
            if not bad.any():
                continue
            lgt, rough, normal, view, uni = ins
            n = normal.shape[0]
            idx = bad.any(dim=0).nonzero().flatten()
            wi2, own2, tab2 = raw_mis(lgt, rough, normal, view, uni)
            torch.cuda.synchronize()
            again = int((~torch.isfinite(tab2)).sum())
            same = bool(torch.equal(torch.nan_to_num(tab2), torch.nan_to_num(tab)) and torch.equal(wi2, wi))
            print('step %d: n=%d, %d points with a non-finite pdf table (samples %s); re-run on the saved inputs: %d '
                  'non-finite, identical to the first run: %s' % (stp, n, idx.numel(), bad.any(dim=1).tolist(), again, same))
            print('   inputs finite: lgt %s rough %s normal %s view %s uniforms %s; |normal| min %.3g max %.3g, rough min %.3g '
                  'max %.3g, uniforms min %.3g max %.9g' % (
                      *[bool(torch.isfinite(t).all()) for t in ins], normal.norm(dim=-1).min().item(),
                      normal.norm(dim=-1).max().item(), rough.min().item(), rough.max().item(), uni.min().item(),
                      uni.max().item()))
            for p_ in idx[:4].tolist():
                print('   point %d: normal %s view %s rough %.4f uniforms %s\n      wi %s\n      own %s tab %s' % (
                    p_, normal[p_].tolist(), view[p_].tolist(), rough.reshape(-1)[p_].item(), uni[p_].tolist(),
                    wi[:, p_].tolist(), own[:, p_].tolist(), tab[:, p_].tolist()))
            shown += 1
            if shown >= 4:
                break


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'step'
    if mode == 'kernel':
        kernel_probe()
    else:
        a = sys.argv[2:]
        step_probe(a[0] if a else 'cfg3', int(a[1]) if len(a) > 1 else 30, int(a[2]) if len(a) > 2 else 3)
