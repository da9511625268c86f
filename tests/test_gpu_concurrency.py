"""Kernels that run BESIDE other streams' work (TrainStep traces the coming batches on side streams while the current
batch's tail runs) must not change anybody's results - round 2's config 3 lost a third of its steps to non-finite
gradients that appeared only under trace prefetch (DESIGN.md "Packed fp32 beside MFMA waves").

Three pins:
  * the instruction form: packed fp32 with op_sel on src1, in inline assembly, beside every evaluator (bit-identical);
  * the mechanism: a canary build of the shading kernels WITH packed-fp32 instructions (libnefii_canary.so: test
    infrastructure, tests/canary/build_canary.py) gives bit-identical results beside every tracer evaluator of the product
    library - on gfx950 it does not when an evaluator leaves room for a foreign wave on its SIMDs (mlp_tile.h, NEFII_CLAIM_SIMD);
  * the symptom: config 3 at full width, 30 steps with three batches of lookahead - no step cancelled, same losses and
    parameters as the serial schedule."""
import ctypes
import os

import pytest
import torch

from nefii_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(name, scene='bowl', seed=0):
    from nefii_amd import conf
    from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
    mc = syn.model_conf(name)
    m = IDRNetwork(conf.from_dict(mc))
    m.load_state_dict(syn.make_state_dict(mc, seed=seed, scene=scene), strict=True)
    m = m.to(DEV)
    m.freeze_geometry()
    m.train()
    return mc, m


def _canary():
    """path of libnefii_canary.so (test infrastructure; built here when it did not travel prebuilt)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('build_canary', os.path.join(ROOT, 'tests', 'canary', 'build_canary.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_canary(verbose=False)


def test_packed_fp32_canary_beside_the_evaluators():
    from nefii_amd import ops
    from nefii_amd.ops import _ptr
    canary = ctypes.CDLL(_canary())
    P, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    canary.nefii_mis_sample.restype = I
    canary.nefii_mis_sample.argtypes = [P, I, P, P, P, P, I64, P, P, P, P]
    _, m512 = _model('conf')
    _, m256 = _model('neus')
    pm512, pm256 = (m.implicit_network.packed(f16x3=True) for m in (m512, m256))
    g = torch.Generator().manual_seed(3)
    n = 114891
    normal = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(DEV)
    view = torch.nn.functional.normalize(torch.randn(n, 3, generator=g) * 0.2 + torch.tensor([0., 0., 1.]), dim=-1).to(DEV)
    rough = (torch.rand(n, generator=g) * 0.8 + 0.15).to(DEV)
    uni = torch.rand(n, 7, generator=g).to(DEV)
    lgt = m512.envmap_material_network.get_lgtSGs().detach().clone().contiguous()
    xs = (torch.randn(1 << 19, 3, generator=g) * 0.45).to(DEV)
    xsmall = (torch.randn(6000, 3, generator=g) * 0.45).to(DEV)

    def victim():
        wi = torch.empty(3, n, 3, device=DEV)
        own = torch.empty(3, n, device=DEV)
        tab = torch.empty(3, n, 3, device=DEV)
        rc = canary.nefii_mis_sample(_ptr(lgt), lgt.shape[0], _ptr(rough), _ptr(normal), _ptr(view), _ptr(uni), n, _ptr(wi),
                                     _ptr(own), _ptr(tab), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return torch.cat([wi.reshape(-1), own.reshape(-1), tab.reshape(-1)])

    loads = {
        '512-wide single pass': lambda: ops.sdf_eval(pm512, xs, coarse=True),
        '512-wide split': lambda: ops.sdf_eval(pm512, xs),
        '256-wide single pass': lambda: ops.sdf_eval(pm256, xs, coarse=True),
        '256-wide split': lambda: ops.sdf_eval(pm256, xs),
        'tracer rounds (all evaluator instances of a 512-wide trace)': None,
    }
    ref = victim()
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(victim()), torch.nan_to_num(ref))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    inp, _ = syn.make_inputs(4096, (800, 800), 1111.0, (0., 0., 2.4), 16, seed=1)
    inp = {k: v.to(DEV) for k, v in inp.items()}
    for name, load in loads.items():
        torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            if load is None:
                for _ in range(4):
                    m512.trace_points(inp)          # eval_kernel16q / 16s instances, 64- and 32-query tiles, as the step runs them
                    ops.sdf_eval(pm512, xsmall)
            else:
                for _ in range(10):
                    load()
        outs = []
        with torch.cuda.stream(sa):
            for _ in range(30):
                outs.append(victim())
        torch.cuda.synchronize()
        bad = [int((torch.nan_to_num(o) != torch.nan_to_num(ref)).sum()) for o in outs]
        assert max(bad) == 0, '%s: %d of %d runs of the packed-fp32 canary differ (worst: %d elements)' % (
            name, sum(b > 0 for b in bad), len(bad), max(bad))


def test_op_sel_canary_beside_the_evaluators():
    """The instruction form itself (tests/canary/pk_forms.hip, inline assembly): a packed-fp32 instruction with op_sel set on
    src1 - both result lanes read src1's HIGH register.  Beside an evaluator that leaves 80 registers of its SIMDs free, lanes
    48-63 of such a wave get ZERO for the low result lane's src1 in 40 of 40 runs (profiles/r03/nan_hunt/13, 17); beside the
    product's evaluators, which claim their SIMDs, every run must be bit-identical to the idle-chip result."""
    from nefii_amd import ops
    from nefii_amd.ops import _ptr
    canary = ctypes.CDLL(_canary())
    P, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    canary.nefii_canary_pk_form.restype = I
    canary.nefii_canary_pk_form.argtypes = [I, P, P, I64, I, P]
    _, m512 = _model('conf')
    _, m256 = _model('neus')
    pm512, pm256 = (m.implicit_network.packed(f16x3=True) for m in (m512, m256))
    g = torch.Generator().manual_seed(3)
    n = 114891
    vin = torch.empty(n, 6)
    vin[:, 0:2] = 0.9 + 0.09 * torch.rand(n, 2, generator=g)
    vin[:, 2:4] = 0.1 * torch.randn(n, 2, generator=g)
    vin[:, 4:6] = torch.randn(n, 2, generator=g)
    vin = vin.to(DEV)
    xs = (torch.randn(1 << 19, 3, generator=g) * 0.45).to(DEV)

    def victim(form):
        out = torch.empty(n, 2, device=DEV)
        assert canary.nefii_canary_pk_form(form, _ptr(vin), _ptr(out), n, 100, torch.cuda.current_stream().cuda_stream) == 0
        return out
    forms = {10: 'v_pk_mul_f32 op_sel:[0,1]', 12: 'v_pk_fma_f32 op_sel:[0,1,0]', 13: 'v_pk_add_f32 op_sel:[0,1]'}
    refs = {f: victim(f) for f in forms}
    torch.cuda.synchronize()
    loads = {'512-wide single pass': lambda: ops.sdf_eval(pm512, xs, coarse=True), '512-wide split': lambda: ops.sdf_eval(pm512, xs),
             '256-wide single pass': lambda: ops.sdf_eval(pm256, xs, coarse=True), '256-wide split': lambda: ops.sdf_eval(pm256, xs),
             'value + gradient': lambda: ops.sdf_value_grad(pm512, xs, want_feat=True)}
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for name, load in loads.items():
        torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            for _ in range(10):
                load()
        outs = []
        with torch.cuda.stream(sa):
            for _ in range(10):
                outs += [(f, victim(f)) for f in forms]
        torch.cuda.synchronize()
        bad = [(forms[f], int((o != refs[f]).sum())) for f, o in outs if not torch.equal(o, refs[f])]
        assert not bad, 'beside %s: %d of %d runs differ, e.g. %s' % (name, len(bad), len(outs), bad[:3])


def test_config3_prefetch_trajectory_equals_the_serial_schedule():
    """BASELINE config 3 (conf.conf at full width, 4096 px x 64 rays, MC direct + indirect, secondary-consistency step every
    10 iterations) for 30 steps: with three batches traced ahead beside the tail no step is cancelled by the NaN guard, and
    losses and parameters follow the serial schedule's (same seeds: the min-SDF draws and the MC uniforms come in the same
    order; the float atomics of the weight-gradient kernels make the two runs differ in the last bits only)."""
    from nefii_amd.training.step import TrainStep
    w = dict(syn.WORKLOADS['cfg3'])
    runs = []
    for lookahead in (0, 3):
        torch.manual_seed(77)
        mc, m = _model(w['model'], w['scene'])
        inp, gt = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1)
        inp = {k: v.to(DEV) for k, v in inp.items()}
        gt = {'rgb': gt.to(DEV)}
        st = TrainStep(m, syn.loss_conf(w['model']), secondary_train_interval=10, secondary_batch_size=1024,
                       num_rays=w['num_rays'])
        nxt = [inp] * lookahead if lookahead else None
        losses = []
        for _ in range(30):
            out, lo = st(inp, gt, nxt)
            losses.append(lo['loss'].detach())
        torch.cuda.synchronize()
        runs.append((torch.stack(losses).cpu(), int(st.nonfinite_steps.item()),
                     {k: v.detach().clone() for k, v in m.state_dict().items()}))
    (l0, bad0, p0), (l1, bad1, p1) = runs
    assert bad0 == 0 and bad1 == 0, (bad0, bad1)
    assert torch.isfinite(l0).all() and torch.isfinite(l1).all()
    assert ((l0 - l1).abs() / l0.abs()).max().item() < 2e-3, (l0, l1)
    assert l1[-5:].mean() < l1[:5].mean()            # and it trains
    for k in p0:
        if p0[k].dtype.is_floating_point and p0[k].numel() > 0:
            d = (p1[k] - p0[k]).norm().item() / (p0[k].norm().item() + 1e-12)
            assert d < 5e-3, (k, d)


def test_a_foreign_reduction_kernel_completes_beside_three_traces():
    """RCCL readiness without RCCL peers.  While TrainStep keeps three traces in flight their evaluators
    claim whole SIMDs (mlp_tile.h NEFII_CLAIM_SIMD_2): a foreign kernel - RCCL's reduction kernels on a real node - only gets
    CUs that host no evaluator workgroup at that moment.  Stand-in for the all-reduce of the 12.95-MB gradient buffer: torch's
    own elementwise add over a buffer of that size (a foreign kernel of a few hundred workgroups, like RCCL's) on a side stream,
    enqueued once the traces are running.  Measured: its latency beside three config-3-sized traces against its latency alone and
    against the traces' own duration.  Asserted: it completes while the traces are still in flight (it is not parked behind them),
    within a bounded multiple of its stand-alone time - i.e. no 'leave N CUs free' mode is needed for world_size > 1; the figures are
    printed for DESIGN.md section 6."""
    import time
    from nefii_amd import ops
    _, m = _model('conf', scene='bowl_trained')
    rt = m.ray_tracer
    w = syn.WORKLOADS['cfg3']
    inp, _ = syn.make_inputs(w['num_pixels'], w['image_hw'], w['focal'], w['cam_pos'], w['num_rays'], seed=1)
    inp = {k: v.to(DEV) for k, v in inp.items()}
    with torch.no_grad():
        m.trace_points(inp)                      # warm-up: packing, calibration of tau / L, allocator
    torch.cuda.synchronize()
    n = 12_950_000 // 4
    buf, other = torch.zeros(n, device=DEV), torch.ones(n, device=DEV)
    side = torch.cuda.Stream()

    def reduction(reps=8):
        with torch.cuda.stream(side):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):                # a ring all-reduce of W ranks runs 2 (W - 1) such passes over 1/W of the buffer each
                buf.add_(other)
            e1.record()
        return e0, e1

    e0, e1 = reduction()
    torch.cuda.synchronize()
    alone_ms = e0.elapsed_time(e1)
    streams = [torch.cuda.Stream() for _ in range(3)]
    ends = []
    t_all0 = torch.cuda.Event(enable_timing=True)
    t_all0.record()
    rt.concurrent = True
    try:
        for st in streams:
            checks = []
            rt.deferred_checks = checks
            with torch.cuda.stream(st):
                st.wait_event(t_all0)
                with torch.no_grad():
                    m.trace_points(inp)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                ends.append(ev)
    finally:
        rt.deferred_checks = None
        rt.concurrent = False
    time.sleep(0.02)                              # the traces are running (each takes ~60-150 ms of evaluator time)
    assert not any(ev.query() for ev in ends), 'the traces finished before the reduction was enqueued: nothing measured'
    e0, e1 = reduction()
    e1.synchronize()
    still_running = sum(0 if ev.query() else 1 for ev in ends)
    beside_ms = e0.elapsed_time(e1)
    torch.cuda.synchronize()
    traces_ms = max(t_all0.elapsed_time(ev) for ev in ends)
    print('[foreign reduction beside three traces] 8 passes over a 12.95-MB buffer: alone %.3f ms, beside three config-3 traces '
          '%.3f ms (x %.1f); the traces took %.1f ms, %d of 3 still in flight when the reduction completed' % (
              alone_ms, beside_ms, beside_ms / alone_ms, traces_ms, still_running))
    assert still_running >= 1, 'the foreign kernel was parked until the traces had finished'
    assert beside_ms < 0.25 * traces_ms, (beside_ms, traces_ms)
    assert float(buf[0].item()) == 16.0 and float(buf[-1].item()) == 16.0
