"""What does the disturbed instruction compute?    The victim form - v_pk_mul_f32 d, x, a op_sel:[0,1];
v_pk_add_f32 x, d, b - four steps per thread (tests/canary/pk_forms.hip form 10, iters = 1), beside the single-pass evaluator
of a build WITHOUT the register claim; every thread whose result differs from the idle-chip one is re-computed on the host
under hypotheses of the kind "in step s the low / high result lane multiplied by a.lo, or by the a.hi of the lane 16 / 32 / 48
below".

    NEFII_LIB_PATH=<libnefii built with -DNEFII_NO_CLAIM> python tools/op_sel_autopsy.py"""
import ctypes
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def simulate(a, b, x, fault=None):
    """float32, one rounding per instruction.  fault = (steps, lanes, value): in those steps (a set) those result lanes (a set
    out of 0 = lo, 1 = hi) multiply by `value` instead of a.hi."""
    x = x.astype(np.float32).copy()
    for s in range(4):
        t = np.array([x[0] * a[1], x[1] * a[1]], dtype=np.float32)
        if fault is not None and s in fault[0]:
            for ln in fault[1]:
                t[ln] = x[ln] * np.float32(fault[2])
        x = (t + b).astype(np.float32)
    return x


def main():
    import torch
    from nefii_amd import build, conf, ops, synthetic as syn
    from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
    from nefii_amd.ops import _ptr
    build.build_canary(verbose=False)
    can = ctypes.CDLL(build.CANARY_OUT)
    P, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    can.nefii_canary_pk_form.restype = I
    can.nefii_canary_pk_form.argtypes = [I, P, P, I64, I, P]
    dev = 'cuda'
    mc = syn.model_conf('conf')
    model = IDRNetwork(conf.from_dict(mc))
    model.load_state_dict(syn.make_state_dict(mc, seed=0, scene='bowl'), strict=True)
    model = model.to(dev)
    model.freeze_geometry()
    pm = model.implicit_network.packed(f16x3=True)
    g = torch.Generator().manual_seed(3)
    n = 1 << 20
    vin = torch.empty(n, 6)
    vin[:, 0:2] = 0.5 + 0.45 * torch.rand(n, 2, generator=g)
    vin[:, 2:4] = torch.randn(n, 2, generator=g)
    vin[:, 4:6] = torch.randn(n, 2, generator=g)
    vin = vin.to(dev)
    xs = (torch.randn(1 << 19, 3, generator=g) * 0.45).to(dev)

    def run():
        out = torch.empty(n, 2, device=dev)
        assert can.nefii_canary_pk_form(10, _ptr(vin), _ptr(out), n, 1, torch.cuda.current_stream().cuda_stream) == 0
        return out
    ref = run()
    torch.cuda.synchronize()
    host = vin.cpu().numpy()
    refh = ref.cpu().numpy()
    chk = np.stack([simulate(host[i, 0:2], host[i, 2:4], host[i, 4:6]) for i in range(2000)])
    print('host model of the four steps equals the idle-chip result on 2000 threads:', bool((chk == refh[:2000]).all()))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sb):
        for _ in range(20):
            ops.sdf_eval(pm, xs, coarse=True)
    outs = []
    with torch.cuda.stream(sa):
        for _ in range(40):
            outs.append(run())
    torch.cuda.synchronize()
    tally, unexplained, total = {}, 0, 0
    lanes = np.zeros(4, dtype=np.int64)
    step_sets = [({k}, 'step %d' % k) for k in range(4)] + [({0, 1, 2, 3}, 'all four steps')]
    lane_sets = [({0}, 'lo'), ({1}, 'hi'), ({0, 1}, 'lo and hi')]
    for o in outs:
        bad = torch.nonzero((o != ref).any(1)).flatten().cpu().numpy()
        oh = o.cpu().numpy()
        for i in bad[:100]:
            total += 1
            lanes[(i % 64) // 16] += 1
            a, b, x = host[i, 0:2], host[i, 2:4], host[i, 4:6]
            values = {'ZERO': 0.0, 'a.lo of the own lane (op_sel ignored)': a[0], 'b.lo': b[0], 'b.hi': b[1]}
            for back in (16, 32, 48):
                values['a.hi of lane - %d' % back] = host[i - back, 1]
                values['a.lo of lane - %d' % back] = host[i - back, 0]
            hit = []
            for (ss, sn), (ls, lnm), (vn, vv) in itertools.product(step_sets, lane_sets, values.items()):
                if (simulate(a, b, x, (ss, ls, vv)) == oh[i]).all():
                    hit.append('%s, result lane %s: multiplied by %s' % (sn, lnm, vn))
            if not hit:
                unexplained += 1
                if unexplained <= 5:
                    print('  unexplained: thread %d a %s b %s x %s -> idle %s got %s' % (i, a, b, x, refh[i], oh[i]))
            for key in hit:
                tally[key] = tally.get(key, 0) + 1.0 / len(hit)
    print('%d wrong threads examined; quarter of the wave (lanes 0-15, 16-31, 32-47, 48-63): %s; not explained by any '
          'hypothesis: %d' % (total, lanes.tolist(), unexplained))
    for k, v in sorted(tally.items(), key=lambda kv: -kv[1]):
        print('  %-80s %.1f' % (k, v))
    print('(the instruction asks for: result lane lo = xlo * ahi, result lane hi = xhi * ahi)')


if __name__ == '__main__':
    main()
