"""Delta-debugging of the packed-fp32 hazard at the ISA level: take the device assembly hipcc emits for
tests/canary/pk_forms.hip, rewrite ONE kernel's instruction stream (wait states after transcendentals, before / after packed
instructions, ...), assemble each variant into its own code object, and run it through hipModuleLaunchKernel beside the
single-pass SDF evaluator of a build WITHOUT the register claim - bit-comparing with the same variant on an idle chip.

    NEFII_LIB_PATH=<libnefii built with -DNEFII_NO_CLAIM> python tools/asm_delta.py [reps=40] [kind=2]

Test tooling: nothing in nefii_amd uses it."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = '/opt/rocm/lib/llvm/bin'
TRANS = r'v_(sin|cos|rsq|rcp|sqrt|exp|log)_f32'


def kernel_span(lines, name):
    a = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
    b = next(i for i in range(a, len(lines)) if 's_endpgm' in lines[i])
    return a, b


def rewrite(lines, name, rule):
    a, b = kernel_span(lines, name)
    out = list(lines[:a])
    body = lines[a:b + 1]
    out += rule(body)
    tail = list(lines[b + 1:])
    k = next(i for i, l in enumerate(tail) if l.strip() == '.amdhsa_kernel ' + name)        # room for the temporaries
    for i in range(k, k + 60):
        if tail[i].strip().startswith('.end_amdhsa_kernel'):
            break
        tail[i] = re.sub(r'(\.amdhsa_next_free_vgpr|\.amdhsa_accum_offset)\s+\d+', r'\1 64', tail[i])
    return out + tail


def insert(body, pattern, text, where):
    out = []
    for l in body:
        hit = re.search(pattern, l) is not None
        if hit and where == 'before':
            out.append('\t' + text)
        out.append(l)
        if hit and where == 'after':
            out.append('\t' + text)
    return out


def _mods(text):
    m = {}
    for key in ('op_sel', 'op_sel_hi', 'neg_lo', 'neg_hi'):
        r = re.search(r'\b%s:\[([01,]+)\]' % key, text)
        m[key] = [int(c) for c in r.group(1).split(',')] if r else None
    return m


def _half(op, hi):
    """Assembly operand for one half (hi = 0 / 1) of a packed source: v[a:b] / s[a:b] -> the register, a constant -> itself."""
    r = re.fullmatch(r'([vs])\[(\d+):(\d+)\]', op)
    if r is None:
        return op
    return '%s%d' % (r.group(1), int(r.group(2)) + hi)


def depack(line, t0='v60', t1='v61'):
    """One v_pk_{mul,add,fma}_f32 / v_pk_mov_b32 as two scalar VOP3 instructions into temporaries + two moves (sources may
    overlap the destination).  Same IEEE operations, same rounding."""
    r = re.match(r'\s+v_pk_(mul_f32|add_f32|fma_f32|mov_b32)\s+(.*)$', line.split(';')[0].rstrip())
    if r is None:
        return [line]
    kind, rest = r.group(1), r.group(2)
    ops_txt = re.split(r'\s+(?=op_sel|neg_lo|neg_hi)', rest, maxsplit=1)
    ops_ = [o.strip() for o in ops_txt[0].split(',')]
    m = _mods(ops_txt[1] if len(ops_txt) > 1 else '')
    dst, srcs = ops_[0], ops_[1:]
    ns = len(srcs)
    sel = m['op_sel'] or [0] * ns
    selh = m['op_sel_hi'] or [1] * ns
    sel, selh = sel + [0] * (ns - len(sel)), selh + [1] * (ns - len(selh))
    nlo, nhi = m['neg_lo'] or [0] * ns, m['neg_hi'] or [0] * ns
    nlo, nhi = nlo + [0] * (ns - len(nlo)), nhi + [0] * (ns - len(nhi))
    out = []
    for t, which, neg in ((t0, sel, nlo), (t1, selh, nhi)):
        a = [('-' if neg[i] else '') + _half(srcs[i], which[i]) for i in range(ns)]
        if kind == 'mov_b32':
            # v_pk_mov_b32 d, s0, s1: d.lo = half(op_sel[0]) of s0, d.hi = half(op_sel_hi[1]... of s1
            src = _half(srcs[0], sel[0]) if t == t0 else _half(srcs[1], sel[1] if m['op_sel'] else 1)
            out.append('\tv_mov_b32_e32 %s, %s' % (t, src))
        else:
            out.append('\tv_%s_e64 %s, %s' % (kind, t, ', '.join(a)))
    out.append('\tv_mov_b32_e32 %s, %s' % (_half(dst, 0), t0))
    out.append('\tv_mov_b32_e32 %s, %s' % (_half(dst, 1), t1))
    return out


def depack_some(body, pick):
    """pick(i, line) -> True: replace the i-th packed instruction of the kernel."""
    out, i = [], 0
    for l in body:
        if re.match(r'\s+v_pk_(mul_f32|add_f32|fma_f32|mov_b32)\s', l):
            out += depack(l) if pick(i, l) else [l]
            i += 1
        else:
            out.append(l)
    return out


RULES = {
    'as compiled': lambda b: b,
    's_nop 7 after every transcendental': lambda b: insert(b, TRANS, 's_nop 7', 'after'),
    '2 x s_nop 7 after every transcendental': lambda b: insert(b, TRANS, 's_nop 7\n\ts_nop 7', 'after'),
    's_nop 3 before every packed instruction': lambda b: insert(b, r'v_pk_\w+_[fb]32', 's_nop 3', 'before'),
    's_nop 3 after every packed instruction': lambda b: insert(b, r'v_pk_\w+_[fb]32', 's_nop 3', 'after'),
    's_nop 1 after every VALU instruction': lambda b: insert(b, r'^\s+v_', 's_nop 1', 'after'),
    'every packed instruction as two scalar ones': lambda b: depack_some(b, lambda i, l: True),
}
if os.environ.get('ASM_DELTA_RULES'):
    exec(open(os.environ['ASM_DELTA_RULES']).read())       # a file that adds to / replaces RULES


def main():
    import torch
    from nefii_amd import conf, ops, synthetic as syn
    from nefii_amd.model.implicit_differentiable_renderer import IDRNetwork
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    kind = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    name = '_Z17pk_natural_kernelILi%dELb0EEvPKfPfli' % kind
    tmp = tempfile.mkdtemp(prefix='asm_delta_')
    base = os.path.join(tmp, 'forms.s')
    nopk = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops'] if os.environ.get('ASM_DELTA_NOPK') else []
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off'] + nopk +
                          ['--cuda-device-only', '-S', os.path.join(ROOT, 'tests', 'canary', 'pk_forms.hip'), '-o', base],
                          stderr=subprocess.DEVNULL)
    lines = open(base).read().split('\n')
    hip = ctypes.CDLL('libamdhip64.so')
    dev = 'cuda'
    mc = syn.model_conf('conf')
    model = IDRNetwork(conf.from_dict(mc))
    model.load_state_dict(syn.make_state_dict(mc, seed=0, scene='bowl'), strict=True)
    model = model.to(dev)
    model.freeze_geometry()
    pm = model.implicit_network.packed(f16x3=True)
    g = torch.Generator().manual_seed(3)
    n = 114891
    vin = torch.empty(n, 6)
    vin[:, 0:2] = 0.9 + 0.09 * torch.rand(n, 2, generator=g)
    vin[:, 2:4] = 0.1 * torch.randn(n, 2, generator=g)
    vin[:, 4:6] = torch.randn(n, 2, generator=g)
    vin = vin.to(dev)
    xs = (torch.randn(1 << 19, 3, generator=g) * 0.45).to(dev)
    iters = int(os.environ.get('PROBE_ITERS', '400'))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    serial = [0]

    def load(variant_lines):
        serial[0] += 1
        src = os.path.join(tmp, 'v%d.s' % serial[0])         # a fresh path per variant: the runtime caches modules by file name
        with open(src, 'w') as f:
            f.write('\n'.join(variant_lines))
        subprocess.check_call([LLVM + '/clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', src,
                               '-o', src + '.o'])
        subprocess.check_call([LLVM + '/ld.lld', '-shared', src + '.o', '-o', src + '.co'])
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipModuleLoad(ctypes.byref(mod), (src + '.co').encode()) == 0
        assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, name.encode()) == 0

        def run():
            out = torch.empty(n, 2, device=dev)
            a = [ctypes.c_void_p(vin.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_int64(n), ctypes.c_int(iters)]
            argv = (ctypes.c_void_p * 4)(*[ctypes.cast(ctypes.pointer(x), ctypes.c_void_p) for x in a])
            rc = hip.hipModuleLaunchKernel(fn, (n + 127) // 128, 1, 1, 128, 1, 1, 0,
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), argv, None)
            assert rc == 0, rc
            return out.reshape(-1)
        return run, mod

    def measure(run, ref, reps_, loaded):
        torch.cuda.synchronize()
        if loaded:
            with torch.cuda.stream(sb):
                for _ in range(max(12, reps_ // 3)):
                    ops.sdf_eval(pm, xs, coarse=True)
        outs = []
        with torch.cuda.stream(sa):
            for _ in range(reps_):
                outs.append(run())
        torch.cuda.synchronize()
        return [int((torch.nan_to_num(o) != torch.nan_to_num(ref)).sum()) for o in outs], outs

    if os.environ.get('ASM_DELTA_DDMIN'):
        return ddmin(lines, name, load, measure, hip)
    print('kernel %s: runs that differ from the idle-chip result of the SAME variant / runs (elements in the worst run)' % name)
    for rname, rule in RULES.items():
        run, _ = load(rewrite(lines, name, rule))
        ref = run()
        torch.cuda.synchronize()
        if rname == 'as compiled':
            ref0 = ref
        row = ['same values as compiled: %s' % torch.equal(torch.nan_to_num(ref), torch.nan_to_num(ref0))]
        for lname, loaded in (('idle', False), ('beside the single-pass evaluator', True)):
            bad, outs = measure(run, ref, reps, loaded)
            row.append('%s: %d/%d (%d)' % (lname, sum(b > 0 for b in bad), len(bad), max(bad)))
            if max(bad) > 0 and os.environ.get('ASM_DELTA_SHOW'):
                o = outs[bad.index(max(bad))]
                d = (o - ref).abs()
                idx = torch.nonzero(d > 0).flatten()
                rel = d[idx] / ref[idx].abs().clamp_min(1e-6)
                thr = torch.unique(idx // 2)
                print('    wrong values %d in %d threads; |diff| median %.3g max %.3g; relative median %.3g; threads (first 24): %s'
                      % (idx.numel(), thr.numel(), d[idx].median().item(), d[idx].max().item(), rel.median().item(),
                         thr[:24].tolist()))
                print('    wave-lane histogram of wrong threads (lane = thread %% 64), 8 bins of 8 lanes: %s; workgroups hit: %d'
                      % (torch.bincount((thr % 64) // 8, minlength=8).tolist(), torch.unique(thr // 128).numel()))
                for t in thr[:6].tolist():
                    print('      thread %d: ref (%.7g, %.7g) got (%.7g, %.7g)' % (t, ref[2 * t].item(), ref[2 * t + 1].item(),
                                                                                 o[2 * t].item(), o[2 * t + 1].item()))
        print('%-52s %s' % (rname, ' | '.join(row)), flush=True)


def vgprs_of(text):
    text = text.split(';')[0]
    regs = set(int(r) for r in re.findall(r'\bv(\d+)\b', text))
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', text):
        regs.update(range(int(a), int(b) + 1))
    return regs


def ddmin(lines, name, load, measure, hip):
    """Shrink the main loop of the kernel (base: RULES[ASM_DELTA_DDMIN]) to a minimal set of instructions that still gives
    different results beside the evaluator and identical ones on an idle chip.  Registers first written inside the loop are
    initialised before it, so that a variant never reads what another kernel left in the register file."""
    import time
    import torch
    base = rewrite(lines, name, RULES[os.environ['ASM_DELTA_DDMIN']])
    a, b = kernel_span(base, name)
    heads = [i for i in range(a, b) if 'Inner Loop Header' in base[i]]
    top = heads[-1]                                                  # the arithmetic loop is the last one
    end = next(i for i in range(top, b) if re.match(r'\s+s_cbranch_scc', base[i]))
    control = re.compile(r'\s+(s_add_i32|s_cmp_eq_u32|s_cbranch)')
    cand = [i for i in range(top + 1, end) if not control.match(base[i]) and base[i].strip() and not base[i].strip().startswith(';')]
    before = set()
    for l in base[a:top]:
        before |= vgprs_of(l)
    inside = set()
    for i in cand:
        inside |= vgprs_of(base[i])
    init = ['\tv_mov_b32_e32 v%d, 1.0' % r for r in sorted(inside - before)]
    print('loop of %d instructions; %d registers initialised before it: %s' % (len(cand), len(init), sorted(inside - before)))

    def variant(keep):
        keep = set(keep)
        out = []
        for i, l in enumerate(base):
            if i == top:
                out += init
            if i in cand_set and i not in keep:
                continue
            out.append(l)
        return out
    cand_set = set(cand)
    tests = [0]
    t_start = time.time()

    def fails(keep):
        tests[0] += 1
        run, mod = load(variant(keep))
        ref = run()
        torch.cuda.synchronize()
        idle, _ = measure(run, ref, 6, False)
        bad, _ = measure(run, ref, 24, True)
        hip.hipModuleUnload(mod)
        verdict = max(idle) == 0 and sum(x > 0 for x in bad) >= 3
        print('  test %3d: %3d instructions kept -> idle %d/6, loaded %d/24 (%d)  %s' % (
            tests[0], len(keep), sum(x > 0 for x in idle), sum(x > 0 for x in bad), max(bad), 'FAILS' if verdict else 'ok'), flush=True)
        return verdict
    assert fails(cand), 'the base does not fail'
    cur, gran = list(cand), 2
    budget = float(os.environ.get('ASM_DELTA_SECONDS', '420'))
    while len(cur) >= 2 and time.time() - t_start < budget:
        chunk = max(1, len(cur) // gran)
        subsets = [cur[k:k + chunk] for k in range(0, len(cur), chunk)]
        reduced = False
        for sub in subsets:                                   # complements first: drop one chunk
            comp = [x for x in cur if x not in set(sub)]
            if comp and fails(comp):
                cur, gran, reduced = comp, max(gran - 1, 2), True
                break
        if not reduced:
            if gran >= len(cur):
                break
            gran = min(len(cur), gran * 2)
    print('minimal loop body (%d instructions, %d tests, %.0f s):' % (len(cur), tests[0], time.time() - t_start))
    for i in cur:
        print(base[i])
    print('initialised before the loop:', ', '.join(x.strip() for x in init))


if __name__ == '__main__':
    main()
