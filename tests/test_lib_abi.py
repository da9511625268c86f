"""CPU-side checks of the C-ABI boundary: the shared library builds for gfx950, loads, and exports every
symbol include/nefii_amd.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'nefii_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nefii_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_loads_and_exports_header_symbols():
    from nefii_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), 'libnefii_hip.so does not export ' + s
        assert s in _lib.SIGNATURES, 'no ctypes signature for ' + s
    assert lib.nefii_abi_version() == _lib.ABI_VERSION


def test_struct_layout_matches_header():
    from nefii_amd import _lib
    # nefii_layer: 4 x int32 + 4 pointers; nefii_mlp: 8 x int32 + 12 layers
    assert ctypes.sizeof(_lib.Layer) == 16 + 5 * 8
    assert ctypes.sizeof(_lib.Mlp) == 40 + 12 * ctypes.sizeof(_lib.Layer)
    assert ctypes.sizeof(_lib.TracerParams) == 80         # 3 floats + 7 int32 + coarse_tau + coarse_cap + minsdf_group + small_round + trace_tier + tier_kappa + tier_gate + minsdf_lipschitz + unread_misses + split_fp8
    assert ctypes.sizeof(_lib.RowBlock) == 32             # 2 pointers + cols + src_row_stride + fill + reserved
    assert ctypes.sizeof(_lib.PackSource) == 48           # 2 pointers + 6 int32 + scale + skip_f32


def test_host_side_argument_checks_need_no_gpu():
    from nefii_amd import _lib
    lib = _lib.lib()
    p = _lib.TracerParams()
    p.n_steps, p.sphere_tracing_iters, p.line_step_iters, p.n_rootfind_steps = 100, 10, 3, 32
    assert lib.nefii_trace_max_rounds(ctypes.byref(p)) == 1 + 10 * 4 + 1 + 11 + 1 + 2    # bisection: 3 levels per round
    p.bisect_levels = 5
    assert lib.nefii_trace_max_rounds(ctypes.byref(p)) == 1 + 10 * 4 + 1 + 7 + 1 + 2
    w0 = lib.nefii_trace_workspace_bytes(4096, ctypes.byref(p))
    p.coarse_tau = 2e-3          # coarse pass: one more round per dense search, a refine list of coarse_cap (default 64) entries per ray
    assert lib.nefii_trace_max_rounds(ctypes.byref(p)) == 1 + 10 * 4 + 1 + 7 + 1 + 2 + 7      # coarse: + chunk, refine x 2, quarter rows x 3, two-stage
    assert lib.nefii_trace_workspace_bytes(4096, ctypes.byref(p)) >= w0 + 4096 * 64 * 4
    p.trace_tier = 1             # tiered sphere tracing: every sphere-tracing evaluation may be repeated in split precision
    assert lib.nefii_trace_max_rounds(ctypes.byref(p)) == 2 * (1 + 10 * 4) + 1 + 7 + 1 + 2 + 7
    p.trace_tier = 0
    w1 = lib.nefii_trace_workspace_bytes(4096, ctypes.byref(p))
    p.minsdf_lipschitz = 1.5     # staged searches: one round more, a list of 76 single samples per ray, the sorted order of the min-SDF draws
    assert lib.nefii_trace_max_rounds(ctypes.byref(p)) == 1 + 10 * 4 + 1 + 7 + 1 + 2 + 7 + 1
    assert lib.nefii_trace_workspace_bytes(4096, ctypes.byref(p)) >= w1 + 4096 * 76 * 4 + 100
    p.minsdf_lipschitz = 0.0
    p.coarse_tau = 0.0
    p.trace_tier = 1             # ... and is the coarse pass's: nothing without it
    assert lib.nefii_trace_max_rounds(ctypes.byref(p)) == 1 + 10 * 4 + 1 + 7 + 1 + 2
    p.trace_tier = 0
    assert lib.nefii_trace_workspace_bytes(4096, ctypes.byref(p)) > 4096 * 100 * 4
    assert lib.nefii_trace_rays(None, None, None, None, None, 0, None, None, None, None, None, None, 0, None, None) == -1
    assert lib.nefii_trace_rays_rounds(None, None, None, None, None, 0, None, None, None, None, None, None, 0, None, 0, 0,
                                       None) == -1
    assert lib.nefii_pack_linear(None, None, 1, 1, 0, 0, 0, 0, 1.0, None, None, None, None) == -1
    assert lib.nefii_assemble_rows(None, 1, None, 0, 1, None) == -1 and lib.nefii_gather_rows(None, 1, None, 1, 1, None) == -1
    blk = (_lib.RowBlock * 1)(_lib.RowBlock(None, None, 0, 0, 0.0, 0))
    assert lib.nefii_assemble_rows(blk, 1, None, 0, 4, None) == -2          # zero columns
    assert lib.nefii_assemble_rows(blk, _lib.MAX_ROW_BLOCKS + 1, None, 0, 4, None) == -1


def test_ops_fail_loudly_without_gpu_tensor():
    import torch
    from nefii_amd import ops
    with pytest.raises(RuntimeError):
        ops.camera_rays(torch.zeros(1, 4, 2), torch.eye(4)[None], torch.eye(4)[None])


def test_padded_width_rule_matches_host():
    from nefii_amd import _lib, ops
    lib = _lib.lib()
    for v in [0, 1, 3, 31, 32, 33, 39, 64, 65, 217, 256, 257, 473, 480, 512]:
        assert lib.nefii_padded_width(v) == ops._pad_hidden(v)
    assert lib.nefii_padded_width(473) == 512 and lib.nefii_padded_width(25) == 32


def test_sdf_stream_size():
    from nefii_amd import _lib, ops, synthetic as syn
    lib = _lib.lib()
    # 512-wide: 16-deep k-steps (either layout); 256-wide: 32-deep k-steps of K padded to 128, 16x16x32 layout only
    for name, hidden, layout, want in [('physg', 512, 0, 8 * (4 + 32 * 3 + 36 + 32 * 3) * 4096),
                                       # 16x16x32 layout: + the K-padded copy of the deep-prefetch 32-query instance
                                       # + the single-pass (coarse) copy: hi fragments only, one 32-deep k-step of the
                                       # wave's 4 (2) feature tiles per 4 KiB (2 KiB) unit, K padded to 128
                                       ('physg', 512, 1, 8 * ((4 + 32 * 3 + 36 + 32 * 3) + (8 + 32 * 3 + 40 + 32 * 3)) * 4096
                                        + 8 * (4 + 16 * 3 + 20 + 16 * 3) * 4096
                                        # + the fifth copy (ABI 15, the "16f" evaluator: correction products on block-scaled fp8):
                                        # eight 4-KiB units per 128-deep chunk of every layer's 128-padded K
                                        + 8 * 8 * (1 + 4 * 3 + 5 + 4 * 3) * 4096),
                                       ('physg', 64, 1, 0),
                                       ('neus', None, 0, 0),
                                       ('neus', None, 1, 8 * (4 + 8 * 3 + 12 + 8 * 3) * 4096 + 8 * (4 + 8 * 3 + 12 + 8 * 3) * 2048)]:
        mc = syn.model_conf(name, hidden=hidden)
        specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
        m = _lib.Mlp()
        m.reserved = layout
        m.n_layers = len(specs)
        for l, s in enumerate(specs):
            m.layer[l].k_x, m.layer[l].k_e, m.layer[l].n_out, m.layer[l].n_pad = s.k_x, s.k_e, s.n_out, s.n_pad
        assert lib.nefii_sdf_stream_bytes(ctypes.byref(m)) == want


def test_value_grad_and_mlp_stream_sizes():
    """The fourth copy of the SDF stream (forward + transposed units for nefii_sdf_value_grad) appears once every layer
    carries transposed fragments, and the radiance / material nets' own stream follows the 64-padded K of their layers
    (sizes only: no kernel runs here)."""
    from nefii_amd import _lib, ops, synthetic as syn
    lib = _lib.lib()

    def descriptor(specs, act, with_bwd):
        m = _lib.Mlp()
        m.n_layers, m.act, m.reserved = len(specs), act, 1
        for l, s in enumerate(specs):
            L = m.layer[l]
            L.k_x, L.k_e, L.n_out, L.n_pad = s.k_x, s.k_e, s.n_out, s.n_pad
            L.w_f16x3, L.bias = 64, 64                  # non-null: only looked at, never read
            L.w_bwd_f16x3 = 64 if with_bwd else None
        return m

    for name, fwd_units, plain in [('physg', 4 + 32 * 3 + 36 + 32 * 3, None), ('neus', 4 + 8 * 3 + 12 + 8 * 3, None)]:
        mc = syn.model_conf(name)
        specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
        m0, m1 = descriptor(specs, ops.ACT_SOFTPLUS100, False), descriptor(specs, ops.ACT_SOFTPLUS100, True)
        for m in (m0, m1):
            m.enc_freqs[0], m.enc_freqs[1], m.enc_freqs[2] = enc[0], -1, -1
        base = lib.nefii_sdf_stream_bytes(ctypes.byref(m0))
        bwd_units = (len(specs) - 2) * (32 if name == 'physg' else 8)
        assert lib.nefii_sdf_stream_bytes(ctypes.byref(m1)) == base + 8 * (fwd_units + bwd_units) * 4096
        m1.act = ops.ACT_RELU                           # the streamed kernel is the softplus nets'
        assert lib.nefii_sdf_stream_bytes(ctypes.byref(m1)) == base
    mc = syn.model_conf('conf')
    F = mc['feature_vector_size']
    specs, enc, head = ops.radiance_specs(mc['rendering_network'], F)
    assert lib.nefii_mlp_stream_bytes(ctypes.byref(descriptor(specs, ops.ACT_RELU, False))) == \
        8 * ((512 + 128) // 16 + 3 * 32) * 4096         # layer 0: 512 features + 96 encoding columns, K 608 -> 640
    specs, enc = ops.material_specs(mc['envmap_material_network'], F, 4)
    assert lib.nefii_mlp_stream_bytes(ctypes.byref(descriptor(specs, ops.ACT_ELU, False))) == \
        8 * ((512 + 64) // 16 + (len(specs) - 2) * 32) * 4096
    specs, enc, head = ops.radiance_specs(syn.model_conf('conf', hidden=64)['rendering_network'], 64)
    assert lib.nefii_mlp_stream_bytes(ctypes.byref(descriptor(specs, ops.ACT_RELU, False))) == 0



def _integration_stub():
    """The ```python block of INTEGRATION.md section B (the binding a maintainer would paste), up to - not including -
    the first line that talks to the library."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    stub = [b for b in blocks if 'class TracerParams' in b]
    assert len(stub) == 1, 'INTEGRATION.md: exactly one ctypes stub expected'
    return stub[0]


def test_integration_md_stub_matches_the_abi():
    """INTEGRATION.md's reference-side binding is executable documentation: its ctypes structs must have the field order,
    types and sizes of nefii_amd/_lib.py (which the GPU suite runs on), its version constant the header's, and it must
    perform the nefii_abi_version() handshake.  (Round 2 shipped a stub that ended at coarse_cap while the header had grown
    minsdf_group / small_round: a pasted copy would have handed the library a short struct.)"""
    from nefii_amd import _lib
    stub = _integration_stub()
    # the struct definitions and the version constant: everything that needs no library handle
    lines, keep = [], False
    for ln in stub.splitlines():
        if ln.startswith('class ') or ln.startswith('NEFII_ABI_VERSION'):
            keep = True
        elif ln and not ln[0].isspace() and not ln.startswith('class '):
            keep = ln.startswith('NEFII_ABI_VERSION')
        if keep:
            lines.append(ln)
    ns = {'ctypes': ctypes}
    exec('\n'.join(lines), ns)
    for name in ('Layer', 'Mlp', 'TracerParams'):
        doc, own = ns[name], getattr(_lib, name)
        assert ctypes.sizeof(doc) == ctypes.sizeof(own), name
        assert [f[0] for f in doc._fields_] == [f[0] for f in own._fields_], name
        for (fn, ft), (_, ot) in zip(doc._fields_, own._fields_):
            assert ctypes.sizeof(ft) == ctypes.sizeof(ot) and getattr(doc, fn).offset == getattr(own, fn).offset, (name, fn)
    header = open(os.path.join(ROOT, 'include', 'nefii_amd.h')).read()
    version = int(re.search(r'#define\s+NEFII_ABI_VERSION\s+(\d+)', header).group(1))
    assert ns['NEFII_ABI_VERSION'] == version == _lib.ABI_VERSION
    assert 'lib.nefii_abi_version() != NEFII_ABI_VERSION' in stub, 'the stub must check the library version'
    # the positional TracerParams(...) call of the stub fills every field
    call = re.search(r'TracerParams\((self\..*?)\)\s*#', stub, flags=re.S).group(1)
    assert len([a for a in call.replace('\n', ' ').split(',') if a.strip()]) == len(_lib.TracerParams._fields_)
    # and the header's struct has the same members in the same order
    body = re.search(r'typedef struct nefii_tracer_params \{(.*?)\} nefii_tracer_params;', header, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    members = [m.strip() for decl in re.findall(r'(?:float|int32_t)\s+([^;]+);', body) for m in decl.split(',')]
    assert members == [f[0] for f in _lib.TracerParams._fields_]
