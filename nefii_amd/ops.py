"""Tensor-level wrappers and autograd glue over the C ABI (include/nefii_amd.h).

PyTorch is plumbing here: it owns device memory, the stream and the autograd graph between the
HIP kernels; every hot op below runs in libnefii_hip.so.  There is no eager fallback - without the
library (or without a GPU tensor) these functions raise.
"""
import ctypes
import os
import math

import torch

from . import _lib
from ._lib import (ACT_ELU, ACT_RELU, ACT_SOFTPLUS100, HEAD_ABS, HEAD_NONE, HEAD_POW2, HEAD_RELU, HEAD_RELU_INIT,
                   HEAD_SIGMOID, HEAD_TANH01, Mlp, TracerParams)


def _round32(v):
    return (v + 31) // 32 * 32


def _pad_hidden(v):
    """nefii_padded_width: hidden/output widths are multiples of 64 once wider than one 32-column tile."""
    return _round32(v) if v <= 32 else (v + 63) // 64 * 64


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('nefii_amd ops need GPU tensors (the hot path has no CPU fallback)')
    if not t.is_contiguous():
        raise RuntimeError('tensor must be contiguous')
    return t.data_ptr()


def _f32(t):
    if t is None:
        return None
    return t.detach().to(torch.float32).contiguous()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw getter is ~20x cheaper than building a
    torch.cuda.Stream object: ~1300 calls per 300 steps of config 1)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class LayerSpec:
    """How one nn.Linear [n_out, k_in] maps onto the kernel's [X | E] input blocks."""

    def __init__(self, n_out, k_in, x_src0=0, x_len=0, e_src0=0, e_len=0, scale=1.0):
        self.n_out, self.k_in = n_out, k_in
        self.x_src0, self.x_len, self.e_src0, self.e_len, self.scale = x_src0, x_len, e_src0, e_len, scale
        self.k_x, self.k_e, self.n_pad = _pad_hidden(x_len), _round32(e_len), _pad_hidden(n_out)


class PackedMLP:
    """Device-resident packed weights + the nefii_mlp descriptor of one fused MLP."""

    def __init__(self, specs, act, head, enc_freqs, feat_width, device, need_bwd=True, f16x3=False, half=False):
        """half: the MLP runs on the fp16-MFMA kernels (nefii_mlp_forward_f16 / _backward_f16 / _wgrad_f16; radiance and
        material networks, never the SDF network) - implies the fp16 fragment sets.  True / 'f16x3': split-precision
        forward, single-pass backward and weight gradients; 'f16': everything in one fp16 pass."""
        assert 1 <= len(specs) <= _lib.MAX_LAYERS
        self.half = half if half in ('f16', 'f16x3') else ('f16x3' if half else False)
        f16x3 = f16x3 or bool(self.half)
        self.specs = specs
        self.act, self.head = act, head
        self.enc_freqs = list(enc_freqs)
        self.feat_width = feat_width
        self.device = device
        self.need_bwd = need_bwd
        self.w_fwd, self.w_bwd, self.bias, self.w_f16, self.w_f16b = [], [], [], [], []
        self.f16x3 = f16x3
        m = Mlp()
        m.n_layers, m.act, m.head, m.feat_width = len(specs), act, head, feat_width
        for i in range(3):
            m.enc_freqs[i] = self.enc_freqs[i]
        for l, s in enumerate(specs):
            k = s.k_x + s.k_e
            # nets on the fp16-MFMA kernels never get their f32 fragments packed (pack(): skip_f32) - they are not allocated
            # and the descriptor carries NULL, so that an f32 entry point called on such a net returns NEFII_E_ARG instead of
            # computing with zero weights
            self.w_fwd.append(None if self.half else torch.zeros(k * s.n_pad, device=device, dtype=torch.float32))
            self.w_bwd.append(torch.zeros(k * s.n_pad, device=device, dtype=torch.float32)
                              if need_bwd and not self.half else None)
            self.bias.append(torch.zeros(s.n_pad, device=device, dtype=torch.float32))
            self.w_f16.append(torch.zeros(2 * k * s.n_pad, device=device, dtype=torch.float16) if f16x3 else None)
            self.w_f16b.append(torch.zeros(2 * k * s.n_pad, device=device, dtype=torch.float16)
                               if f16x3 and need_bwd else None)
            L = m.layer[l]
            L.k_x, L.k_e, L.n_out, L.n_pad = s.k_x, s.k_e, s.n_out, s.n_pad
            L.w_fwd = self.w_fwd[l].data_ptr() if self.w_fwd[l] is not None else None
            L.w_bwd = self.w_bwd[l].data_ptr() if self.w_bwd[l] is not None else None
            L.bias = self.bias[l].data_ptr()
            L.w_f16x3 = self.w_f16[l].data_ptr() if f16x3 else None
            L.w_bwd_f16x3 = self.w_f16b[l].data_ptr() if self.w_f16b[l] is not None else None
        self.struct = m
        # SDF nets of the qualifying shape also keep their hidden layers as one fragment stream per wave (the
        # pipelined tile evaluator of the tracer); other nets leave w_stream NULL and run on the generic kernel
        self.w_stream = None
        if f16x3 and not self.half and torch.device(device).type == 'cuda':
            # stream layout / matrix instruction of the pipelined evaluator: 0 = 32x32x16 (512-wide nets only),
            # 1 = 16x16x32 (512- and 256-wide nets)
            m.reserved = int(os.environ.get('NEFII_STREAM_LAYOUT', '1'))
            nbytes = _lib.lib().nefii_sdf_stream_bytes(ctypes.byref(m))
            if nbytes and feat_width == 0 and self.enc_freqs[1] < 0 and self.enc_freqs[2] < 0:
                self.w_stream = torch.zeros(nbytes // 2, device=device, dtype=torch.float16)
                m.w_stream = self.w_stream.data_ptr()
            else:
                m.reserved = 0
        # radiance / material nets with 512-wide hidden layers: the split-precision forward reads the hidden layers as a
        # fragment stream too (nefii_mlp_forward_f16 on 48- / 64-row tiles); re-packed with the weights every pack()
        self.mlp_stream = False
        if self.half == 'f16x3' and torch.device(device).type == 'cuda':
            nbytes = _lib.lib().nefii_mlp_stream_bytes(ctypes.byref(m))
            if nbytes:
                self.w_stream = torch.zeros(nbytes // 2, device=device, dtype=torch.float16)
                m.w_stream = self.w_stream.data_ptr()
                self.mlp_stream = True
        self.hidden_stride = max(s.n_pad for s in specs)
        self.packed_version = None

    @property
    def n_layers(self):
        return len(self.specs)

    @property
    def in_width(self):
        return self.specs[0].k_in

    def pack(self, weights, biases):
        """weights[l]: effective [n_out, k_in] fp32 GPU tensors (after weight-norm), biases[l]: [n_out]."""
        lib = _lib.lib()
        st = _stream()
        # every layer and every form in ONE launch (nefii_pack_mlp); the fp16-MFMA kernels (self.half) never read the
        # f32 fragments, so those are skipped there
        src = (_lib.PackSource * len(self.specs))()
        keep = []                               # converted copies must outlive the launch call
        for l, s in enumerate(self.specs):
            w = _f32(weights[l])
            b = _f32(biases[l])
            keep += [w, b]
            assert tuple(w.shape) == (s.n_out, s.k_in), (tuple(w.shape), s.n_out, s.k_in)
            e = src[l]
            e.W, e.bias = _ptr(w), _ptr(b)
            e.n_out, e.k_in, e.x_src0, e.x_len, e.e_src0, e.e_len = s.n_out, s.k_in, s.x_src0, s.x_len, s.e_src0, s.e_len
            e.scale, e.skip_f32 = s.scale, 1 if self.half else 0
        _lib.check(lib.nefii_pack_mlp(ctypes.byref(self.struct), src, st), 'nefii_pack_mlp')
        if self.mlp_stream:
            _lib.check(lib.nefii_pack_mlp_stream(ctypes.byref(self.struct), _ptr(self.w_stream), st),
                       'nefii_pack_mlp_stream')
        elif self.w_stream is not None:
            _lib.check(lib.nefii_pack_sdf_stream(ctypes.byref(self.struct), _ptr(self.w_stream), st),
                       'nefii_pack_sdf_stream')


def mlp_precision():
    """Arithmetic of the radiance / material MLPs (NEFII_MLP_PRECISION): 'f16x3' (default) - fp16 MFMA tiles: forward in
    split precision (hi/lo pairs, fp32-class accuracy), backward and weight gradients in one fp16 pass with a
    power-of-two gradient scale; 'f16' - the forward in one fp16 pass too (measured 1.3e-3 relative L2 on config 3's RGB:
    above the north-star bar, for measurement only); 'f32' - the f32-input MFMA kernels (bit-exact fp32 fma chains)."""
    return os.environ.get('NEFII_MLP_PRECISION', 'f16x3')


def h16_supported(pm):
    """The net trains with its stash and dz kept in halves (nefii_mlp_*_f16h: streamed kernels forward and backward).
    NEFII_MLP_H16=0 keeps the fp32 stash (A/B measurements)."""
    if pm.half != 'f16x3' or os.environ.get('NEFII_MLP_H16', '1') == '0':
        return False
    if getattr(pm, '_h16', None) is None:        # a property of the net's shape and buffers, not of its weights
        pm._h16 = bool(_lib.lib().nefii_mlp_h16_supported(ctypes.byref(pm.struct)))
    return pm._h16


def param_list(module):
    """list(module.parameters()), gathered once per module: the Parameter OBJECTS of the networks here are fixed at
    construction (weight_norm included; load_state_dict / .to() / optimizers write in place), and Module.parameters() walks
    the module tree on every call - the per-step version / requires_grad checks cost 0.09 ms of config 1's 0.81-ms host step."""
    plist = module.__dict__.get('_nefii_plist')
    if plist is not None:
        # ... but nothing stops a caller from REPLACING them (load_state_dict(assign=True), remove_weight_norm, to_empty):
        # the version / requires_grad checks would then watch dead objects and serve stale packed weights.  Cheap guard on
        # every call - the first parameter is still the first (next() stops at the first leaf) - and a full identity
        # check every 256th.
        n = module.__dict__['_nefii_plist_calls'] = module.__dict__.get('_nefii_plist_calls', 0) + 1
        first = next(module.parameters(), None)
        if (plist[0] if plist else None) is not first:
            plist = None
        elif n % 256 == 0:
            cur = list(module.parameters())
            if len(cur) != len(plist) or any(a is not b for a, b in zip(cur, plist)):
                plist = None
    if plist is None:
        plist = module.__dict__['_nefii_plist'] = list(module.parameters())
    return plist


class HalfStash:
    """What nefii_mlp_forward_f16h leaves for the backward pass: h [n_layers - 1, n, stride] halves (16 x the hidden
    activations), z_last [n, 8] floats (pre-activations of the head), x0 [n, nefii_mlp_x0_width] halves (16 x layer 0's
    input in the kernel's column order: padded features, then the encodings)."""

    def __init__(self, h, z_last, x0):
        self.h, self.z_last, self.x0 = h, z_last, x0


def mlp_forward(pm, in_a, in_b, in_c, feat, want_hidden=False, want_stash=False, h16=None):
    """(out, last hidden or None, stash or None).  The stash is a HalfStash where the net supports it (h16=False: the fp32
    [n_layers, n, stride] tensor regardless), else fp32."""
    lib = _lib.lib()
    n = in_a.shape[0]
    dev = in_a.device
    n_out = pm.specs[-1].n_out
    out = torch.empty(n, n_out, device=dev, dtype=torch.float32)
    hidden = None
    if want_hidden:
        hidden = torch.empty(n, pm.specs[-2].n_out, device=dev, dtype=torch.float32)
    stash = None
    if want_stash and n > 0 and h16 is not False and h16_supported(pm):
        stash = HalfStash(torch.empty(pm.n_layers - 1, n, pm.hidden_stride, device=dev, dtype=torch.float16),
                          torch.empty(n, 8, device=dev, dtype=torch.float32),
                          torch.empty(n, lib.nefii_mlp_x0_width(ctypes.byref(pm.struct)), device=dev, dtype=torch.float16))
        _lib.check(lib.nefii_mlp_forward_f16h(ctypes.byref(pm.struct), _ptr(in_a), _ptr(in_b), _ptr(in_c), _ptr(feat), n,
                                              _ptr(out), n_out, _ptr(hidden), hidden.shape[1] if hidden is not None else 0,
                                              _ptr(stash.h), pm.hidden_stride, _ptr(stash.z_last), _ptr(stash.x0), _stream()),
                   'nefii_mlp_forward_f16h')
        return out, hidden, stash
    if want_stash:
        stash = torch.empty(pm.n_layers, n, pm.hidden_stride, device=dev, dtype=torch.float32)
    if n > 0:
        args = (ctypes.byref(pm.struct), _ptr(in_a), _ptr(in_b), _ptr(in_c), _ptr(feat), n, _ptr(out), n_out, _ptr(hidden),
                hidden.shape[1] if hidden is not None else 0, _ptr(stash), pm.hidden_stride)
        if pm.half:
            _lib.check(lib.nefii_mlp_forward_f16(*args, 1 if pm.half == 'f16' else 0, _stream()), 'nefii_mlp_forward_f16')
        else:
            _lib.check(lib.nefii_mlp_forward(*args, _stream()), 'nefii_mlp_forward')
    return out, hidden, stash


def mlp_backward(pm, d_out, stash, gscale=None):
    """gscale (pm.half): device scalar from mlp_grad_scale(d_out)"""
    lib = _lib.lib()
    n = d_out.shape[0]
    if isinstance(stash, HalfStash):        # dz comes back in halves too: gscale x dz_l
        dz = torch.empty(pm.n_layers, n, pm.hidden_stride, device=d_out.device, dtype=torch.float16)
        _lib.check(lib.nefii_mlp_backward_f16h(ctypes.byref(pm.struct), _ptr(d_out), d_out.shape[1], _ptr(stash.h),
                                               pm.hidden_stride, _ptr(stash.z_last), n, _ptr(dz), pm.hidden_stride,
                                               _ptr(gscale), _stream()), 'nefii_mlp_backward_f16h')
        return dz
    dz = torch.empty(pm.n_layers, n, pm.hidden_stride, device=d_out.device, dtype=torch.float32)
    if n > 0:
        if pm.half:
            _lib.check(lib.nefii_mlp_backward_f16(ctypes.byref(pm.struct), _ptr(d_out), d_out.shape[1], _ptr(stash),
                                                  pm.hidden_stride, n, _ptr(dz), pm.hidden_stride, _ptr(gscale), _stream()),
                       'nefii_mlp_backward_f16')
        else:
            _lib.check(lib.nefii_mlp_backward(ctypes.byref(pm.struct), _ptr(d_out), d_out.shape[1], _ptr(stash),
                                              pm.hidden_stride, n, _ptr(dz), pm.hidden_stride, _stream()),
                       'nefii_mlp_backward')
    return dz


def mlp_grad_scale(d_out):
    """power-of-two scale that brings max |d_out| to ~256 (device scalar; the fp16 backward GEMMs carry dz x scale)"""
    s = torch.empty(1, device=d_out.device, dtype=torch.float32)
    _lib.check(_lib.lib().nefii_mlp_grad_scale(_ptr(d_out), d_out.numel(), _ptr(s), _stream()), 'nefii_mlp_grad_scale')
    return s


def encode_inputs(pm, in_a, in_b, in_c, feat):
    lib = _lib.lib()
    n = in_a.shape[0]
    out = torch.empty(n, pm.in_width, device=in_a.device, dtype=torch.float32)
    if n > 0:
        _lib.check(lib.nefii_encode_inputs(ctypes.byref(pm.struct), _ptr(in_a), _ptr(in_b), _ptr(in_c), _ptr(feat), n,
                                           _ptr(out), pm.in_width, _stream()), 'nefii_encode_inputs')
    return out


def _stash_tensors(stash, like):
    if stash is None:
        return (like.new_empty(0),)
    return (stash.h, stash.z_last, stash.x0) if isinstance(stash, HalfStash) else (stash,)


class FusedMLPFn(torch.autograd.Function):
    """y = MLP(PE(a), PE(b), PE(c), feat); differentiable wrt the layer weights and biases only
    (the raw inputs come from frozen geometry: IDRNetwork.freeze_geometry, Step-2; in the Step-1 geometry fit the inputs
    are sample positions)."""

    @staticmethod
    def forward(ctx, pm, in_a, in_b, in_c, feat, *wb):
        L = pm.n_layers
        ws, bs = wb[:L], wb[L:]
        pm.pack(ws, bs)
        need = any(t.requires_grad for t in wb)
        out, _, stash = mlp_forward(pm, in_a, in_b, in_c, feat, want_stash=need)
        ctx.pm = pm
        ctx.save_for_backward(in_a, in_b if in_b is not None else in_a.new_empty(0),
                              in_c if in_c is not None else in_a.new_empty(0),
                              feat if feat is not None else in_a.new_empty(0), *_stash_tensors(stash, in_a))
        ctx.has = (in_b is not None, in_c is not None, feat is not None)
        # an output that nothing differentiable reads (physg.conf weights the radiance colour with 0 and the loss detaches it)
        # may still be reachable in the autograd graph through a node it shares with other outputs (AssembleRowsFn): its
        # backward then arrives with None and must not run on materialised zeros
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, d_out):
        pm = ctx.pm
        if d_out is None:
            return (None,) * (5 + 2 * pm.n_layers)
        in_a, in_b, in_c, feat, *st = ctx.saved_tensors
        stash = HalfStash(*st) if len(st) == 3 else st[0]
        h16 = isinstance(stash, HalfStash)
        in_b = in_b if ctx.has[0] else None
        in_c = in_c if ctx.has[1] else None
        feat = feat if ctx.has[2] else None
        L = pm.n_layers
        d_out = d_out.contiguous()
        gscale = mlp_grad_scale(d_out) if pm.half else None
        dz = mlp_backward(pm, d_out, stash, gscale)
        lib = _lib.lib()
        n = d_out.shape[0]
        gw, gb = [], []
        if h16 and n > 0 and os.environ.get('NEFII_WGRAD_BATCH', '1') != '0':
            # every layer's weight gradient in one zero-fill + one launch per kernel form (nefii_mlp_wgrad_f16h_batch): the
            # per-layer calls below are 2 L dependent launches in the step's tail
            s0 = pm.specs[0]
            k_img = stash.x0.shape[1]
            items = (_lib.WgradItem * L)()
            outs = []
            for l, s in enumerate(pm.specs):
                k = k_img if l == 0 else s.k_in
                g = torch.empty(s.n_out, k, device=d_out.device, dtype=torch.float32)
                b = torch.empty(s.n_out, device=d_out.device, dtype=torch.float32)
                xin, xs = (stash.x0, k_img) if l == 0 else (stash.h[l - 1], pm.hidden_stride)
                items[l] = _lib.WgradItem(_ptr(dz[l]), _ptr(xin), _ptr(g), _ptr(b), pm.hidden_stride, xs, 1, s.n_out, k, s.scale)
                outs.append((g, b))
            _lib.check(lib.nefii_mlp_wgrad_f16h_batch(items, L, n, _ptr(gscale), _stream()), 'nefii_mlp_wgrad_f16h_batch')
            g_img = outs[0][0]      # layer 0 comes in the image's column order (features | encodings): back to the Linear's
            kx = lib.nefii_padded_width(s0.x_len)
            parts = sorted([(s0.x_src0, g_img[:, :s0.x_len]), (s0.e_src0, g_img[:, kx:kx + s0.e_len])], key=lambda t: t[0])
            gw = [torch.cat([t[1] for t in parts if t[1].shape[1]], dim=1)] + [g for g, _ in outs[1:]]
            gb = [b for _, b in outs]
            return (None, None, None, None, None) + tuple(gw) + tuple(gb)
        if h16:     # layer 0's input left by the forward: no encoded matrix to rebuild, dW0 comes in the image's column order
            s0 = pm.specs[0]
            k_img = stash.x0.shape[1]
            g_img = torch.empty(s0.n_out, k_img, device=d_out.device, dtype=torch.float32)
            b = torch.empty(s0.n_out, device=d_out.device, dtype=torch.float32)
            _lib.check(lib.nefii_mlp_wgrad_f16h(_ptr(dz[0]), pm.hidden_stride, _ptr(stash.x0), k_img, 1, n, s0.n_out, k_img,
                                                s0.scale, _ptr(gscale), _ptr(g_img), _ptr(b), _stream()), 'nefii_mlp_wgrad_f16h')
            kx = lib.nefii_padded_width(s0.x_len)
            parts = sorted([(s0.x_src0, g_img[:, :s0.x_len]), (s0.e_src0, g_img[:, kx:kx + s0.e_len])], key=lambda t: t[0])
            gw.append(torch.cat([t[1] for t in parts if t[1].shape[1]], dim=1))
            gb.append(b)
        else:
            x0 = encode_inputs(pm, in_a, in_b, in_c, feat)
        for l, s in enumerate(pm.specs):
            if l == 0 and h16:
                continue
            if l == 0:
                xin, xs = x0, x0.shape[1]
            elif h16:
                xin, xs = stash.h[l - 1], pm.hidden_stride
            elif s.e_len:
                # skip layer: its input is [previous activations | encoded network input] (the 1/sqrt(2) is s.scale)
                # in the Linear's column order (x_src0 / e_src0 are column offsets into its weight)
                blocks = [stash[l - 1][:, :s.x_len], x0[:, :s.e_len]]
                xin = torch.cat(blocks if s.x_src0 < s.e_src0 else blocks[::-1], dim=1).contiguous()
                xs = xin.shape[1]
            else:
                xin, xs = stash[l - 1], pm.hidden_stride
            g = torch.empty(s.n_out, s.k_in, device=d_out.device, dtype=torch.float32)
            b = torch.empty(s.n_out, device=d_out.device, dtype=torch.float32)
            if h16:
                _lib.check(lib.nefii_mlp_wgrad_f16h(_ptr(dz[l]), pm.hidden_stride, _ptr(xin), xs, 1, n, s.n_out, s.k_in, s.scale,
                                                    _ptr(gscale), _ptr(g), _ptr(b), _stream()), 'nefii_mlp_wgrad_f16h')
            elif pm.half:
                _lib.check(lib.nefii_mlp_wgrad_f16(_ptr(dz[l]), pm.hidden_stride, _ptr(xin), xs, n, s.n_out, s.k_in, s.scale,
                                                   _ptr(gscale), _ptr(g), _ptr(b), _stream()), 'nefii_mlp_wgrad_f16')
            else:
                _lib.check(lib.nefii_mlp_wgrad(_ptr(dz[l]), pm.hidden_stride, _ptr(xin), xs, n, s.n_out, s.k_in, s.scale,
                                               _ptr(g), _ptr(b), _stream()), 'nefii_mlp_wgrad')
            gw.append(g)
            gb.append(b)
        return (None, None, None, None, None) + tuple(gw) + tuple(gb)


class FusedMLPHiddenFn(torch.autograd.Function):
    """FusedMLPFn that also returns the last hidden activation (ImplicitNetwork with use_last_as_f) as a
    non-differentiable output: the Step-1 fit consumes the SDF column only."""

    @staticmethod
    def forward(ctx, pm, in_a, *wb):
        L = pm.n_layers
        pm.pack(wb[:L], wb[L:])
        need = any(t.requires_grad for t in wb)
        out, hidden, stash = mlp_forward(pm, in_a, None, None, None, want_hidden=True, want_stash=need)
        ctx.pm = pm
        empty = in_a.new_empty(0)
        ctx.save_for_backward(in_a, empty, empty, empty, *_stash_tensors(stash, in_a))
        ctx.has = (False, False, False)
        ctx.mark_non_differentiable(hidden)
        ctx.set_materialize_grads(False)
        return out, hidden

    @staticmethod
    def backward(ctx, d_out, _d_hidden):
        g = FusedMLPFn.backward(ctx, d_out)
        return g[:2] + g[5:]


def sdf_value_grad(pm, x, want_feat=False):
    """(sdf_out [n, n_out_last], feature [n, hidden] or None, d sdf/dx [n,3])."""
    lib = _lib.lib()
    n = x.shape[0]
    dev = x.device
    n_out = pm.specs[-1].n_out
    out = torch.empty(n, n_out, device=dev, dtype=torch.float32)
    grad = torch.empty(n, 3, device=dev, dtype=torch.float32)
    feat = torch.empty(n, pm.specs[-2].n_out, device=dev, dtype=torch.float32) if want_feat else None
    if n > 0:
        nbytes = lib.nefii_sdf_value_grad_workspace_bytes(ctypes.byref(pm.struct), n)
        ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        _lib.check(lib.nefii_sdf_value_grad(ctypes.byref(pm.struct), _ptr(x), n, _ptr(out), n_out, _ptr(feat),
                                            feat.shape[1] if feat is not None else 0, _ptr(grad), _ptr(ws), _stream()),
                   'nefii_sdf_value_grad')
    return out, feat, grad


def sdf_eval(pm, x, coarse=False, fp8=False):
    """implicit_network(x)[:, 0] with the tracer's split-precision tile evaluator (needs PackedMLP(f16x3=True));
    coarse=True: with its single-pass (one fp16 MFMA per product) evaluator instead; fp8=True: with the "16f" evaluator
    (correction products on block-scaled fp8, nefii_tracer_params.split_fp8)."""
    lib = _lib.lib()
    x = x.contiguous()
    n = x.shape[0]
    out = torch.empty(n, device=x.device, dtype=torch.float32)
    if n > 0:
        fn, name = (lib.nefii_sdf_eval_coarse, 'nefii_sdf_eval_coarse') if coarse else (lib.nefii_sdf_eval, 'nefii_sdf_eval')
        if fp8:
            fn, name = lib.nefii_sdf_eval_fp8corr, 'nefii_sdf_eval_fp8corr'
        _lib.check(fn(ctypes.byref(pm.struct), _ptr(x), n, _ptr(out), _stream()), name)
    return out


def fp8corr_supported(pm):
    return bool(pm.f16x3 and pm.w_stream is not None and _lib.lib().nefii_sdf_fp8corr_supported(ctypes.byref(pm.struct)))


def coarse_supported(pm):
    return bool(pm.f16x3 and pm.w_stream is not None and _lib.lib().nefii_sdf_coarse_supported(ctypes.byref(pm.struct)))


COARSE_TAU_SAFETY = 3.0


def calibrate_coarse_tau(pm, radius=1.0, n=65536, safety=COARSE_TAU_SAFETY, seed=0):
    """Error bound of the tracer's coarse pass for THIS network: `safety` x the largest |single-pass - split| SDF value
    over n points drawn uniformly in the bounding sphere (where the tracer samples), at least 1e-4.  0.0 when the net
    has no single-pass stream.  One host sync; callers cache it per packed weight version (geometry is frozen)."""
    if not coarse_supported(pm):
        return 0.0
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, generator=g)
    x = x / x.norm(dim=1, keepdim=True) * (torch.rand(n, 1, generator=g) ** (1.0 / 3.0)) * (radius * 1.02)
    x = x.to(pm.device)
    err = (sdf_eval(pm, x, coarse=True) - sdf_eval(pm, x)).abs().max().item()
    if not math.isfinite(err):
        return 0.0
    return max(1e-4, safety * err)


LIPSCHITZ_SAFETY = 1.5


def calibrate_lipschitz(grad_fn, device, radius=1.0, n=65536, safety=LIPSCHITZ_SAFETY, seed=1, search_rounds=4):
    """Bound on |sdf(p) - sdf(q)| / |p - q| inside the bounding sphere for THIS network, for the tracer's staged searches
    (nefii_tracer_params.minsdf_lipschitz): `safety` x the largest |grad sdf| found, at least 1.  Found by n points drawn
    uniformly in the sphere, then `search_rounds` rounds of local search around the 256 steepest points so far (255 Gaussian
    perturbations each, radius shrinking from 0.04: the steep spots of a softplus-100 network are small - the search typically
    raises the random sample's maximum by a few percent).  grad_fn: points [n, 3] -> gradients [n, 3] (ImplicitNetwork.gradient).
    Like coarse_tau a MEASURED bound, not a proven one; the tracer audits it (counter column 12).  One host sync; cached per
    packed weight version."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, generator=g)
    x = x / x.norm(dim=1, keepdim=True) * (torch.rand(n, 1, generator=g) ** (1.0 / 3.0)) * (radius * 1.02)
    x = x.to(device)
    gn = grad_fn(x).reshape(-1, 3).norm(dim=1)
    sigma = 0.04 * radius
    for _ in range(search_rounds):
        top = torch.topk(gn, min(256, gn.numel())).indices
        seeds, best = x[top], gn[top]
        pert = torch.randn(seeds.shape[0], 255, 3, generator=g).to(device) * sigma
        cand = (seeds.unsqueeze(1) + pert).reshape(-1, 3)
        rn = cand.norm(dim=1, keepdim=True)
        cand = torch.where(rn > radius * 1.02, cand * (radius * 1.02 / rn), cand)        # stay inside the sphere
        x = torch.cat([seeds, cand])
        gn = torch.cat([best, grad_fn(cand).reshape(-1, 3).norm(dim=1)])
        sigma *= 0.5
    gmax = gn.max().item()
    if not math.isfinite(gmax):
        return 0.0
    return max(1.0, safety * gmax)


def algorithmic_evals(counters, n_steps):
    """SDF evaluations the reference's recurrences need for the rounds in `counters` [..., rounds, 14] (_lib.TRACE_COUNTERS
    columns; what frac_credited credits): singles (0) + tiered singles taken (9 - 10) + n_steps per dense search entered (6)
    + bisection steps consumed (3)."""
    c = counters.long()
    return c[..., 0] + c[..., 9] - c[..., 10] + c[..., 6] * n_steps + c[..., 3]


def executed_evals(counters, n_steps, tri_nodes=None):
    """(split-precision evaluations, coarse single-pass evaluations) actually executed (tri_nodes: unused, the tracer
    counts its speculative bisection evaluations itself)."""
    c = counters.long()
    return c[..., 0] + c[..., 1] * n_steps + c[..., 7] + c[..., 4], c[..., 5] * ((n_steps + 3) // 4) + c[..., 9] + c[..., 11]


PRECISIONS = {'f32': 0, 'f16x3': 1, 'f16x3w': 2}


def make_tracer_params(cfg, training, precision='f32', bisect_levels=3, coarse_tau=0.0, coarse_cap=0, minsdf_group=0,
                       small_round=0, trace_tier=0, tier_kappa=0.0, tier_gate=0.0, minsdf_lipschitz=0.0, unread_misses=0,
                       split_fp8=0):
    p = TracerParams()
    p.split_fp8 = 1 if split_fp8 else 0
    p.unread_misses = 1 if unread_misses else 0
    p.minsdf_lipschitz = float(minsdf_lipschitz) if coarse_tau > 0.0 else 0.0
    p.trace_tier = 1 if (trace_tier and coarse_tau > 0.0) else 0
    p.tier_kappa = float(tier_kappa)
    p.tier_gate = float(tier_gate)
    p.minsdf_group = int(minsdf_group)
    p.small_round = int(small_round)
    p.precision = PRECISIONS[precision]
    p.bisect_levels = bisect_levels
    p.coarse_tau = float(coarse_tau)
    p.coarse_cap = int(coarse_cap)
    p.object_bounding_sphere = cfg.get('object_bounding_sphere', 1.0)
    p.sdf_threshold = cfg.get('sdf_threshold', 5.0e-5)
    p.line_search_step = cfg.get('line_search_step', 0.5)
    p.line_step_iters = cfg.get('line_step_iters', 1)
    p.sphere_tracing_iters = cfg.get('sphere_tracing_iters', 10)
    p.n_steps = cfg.get('n_steps', 100)
    p.n_rootfind_steps = cfg.get('n_rootfind_steps', 8)
    p.training = 1 if training else 0
    return p


class TraceRounds:
    """Adaptive prefix of the tracer's rounds.  Every round after the last one that emitted a query is an empty
    launch pair (~12 us); callers that synchronise right after the trace anyway (the renderer compacts the hits)
    run rounds [0, guess), read three counters and continue only if a ray is still waiting - which the guess,
    taken from the previous call, makes rare.  Results never depend on the guess."""

    def __init__(self):
        self.guess = None


_TRACE_STREAMS = {}
_WORK = [0, 1, 2, 4, 5, 9, 11]   # counter columns that mean "a ray still waits for an evaluation"


_SIDE_STREAMS = {}


def side_stream(dev):
    """One process-wide side stream per device for work that may run beside the caller's serial chain (a detached radiance
    forward, model/implicit_differentiable_renderer.py:get_rbg_value)."""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return st


def _trace_streams(dev, n):
    pool = _TRACE_STREAMS.setdefault(dev, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


AUDIT_COLUMN = 8     # float bits (max); so is LIP_AUDIT_COLUMN; every other column is an additive count
LIP_AUDIT_COLUMN = 12


def sum_counters(a, b=None):
    """Counters of several stream groups (a: [groups, rounds, C] -> [rounds, C]) or of two traces (a + b, [rounds, C]):
    the counts add, column 8 - the bits of a non-negative float, the audit maximum - takes the maximum (the bit patterns
    of non-negative floats order like the floats)."""
    if b is None:
        out = a.sum(dim=0)
        for col in (AUDIT_COLUMN, LIP_AUDIT_COLUMN):
            out[..., col] = a[..., col].max(dim=0).values
        return out
    out = a + b
    for col in (AUDIT_COLUMN, LIP_AUDIT_COLUMN):
        out[..., col] = torch.maximum(a[..., col], b[..., col])
    return out


def _lip_audit_of(host_counters):
    """Largest amount by which a second-stage depth of a staged min-SDF search fell below the lower bound the claimed
    Lipschitz constant gave it (counter column 12: float bits; 0 = the bound held wherever it was checked)."""
    col = host_counters[..., LIP_AUDIT_COLUMN].contiguous().view(torch.float32)
    return float(col.max()) if col.numel() else 0.0


def _audit_of(host_counters):
    """Largest |coarse - split| the tracer saw among the coarse samples it re-evaluated (counter column 8: float bits)."""
    col = host_counters[..., 8].contiguous().view(torch.float32)
    return float(col.max()) if col.numel() else 0.0


def trace_rays(pm_sdf, params, origins, dirs, object_mask, lin_steps, minsdf_steps=None, want_counters=False,
               rounds_state=None, groups=1, deferred=None, audit=None, keep_workspace=None):
    """RayTracing.forward for per-ray origins.  Returns points [n,3], hit (bool [n]), dists [n] (+ counters).

    groups > 1: the rays are cut into that many contiguous chunks that run their rounds on separate HIP streams
    (nefii_trace_rays_groups).  Rays are independent, so the results are bit-identical; the latency-bound rounds of a
    small batch (fewer 64-query tiles than CUs, one tile time per round regardless) of different chunks then overlap
    on the chip.  On MI355X the dense rounds lose as much as that gains (RayTracing.stream_groups), so it is off by
    default.

    audit (a callable, with rounds_state): called with the largest |coarse - split| of this trace's refined samples whenever
    the counters reach the host (the online check of nefii_tracer_params.coarse_tau: ImplicitNetwork.note_coarse_audit).

    keep_workspace (a list): receives (workspace tensor, first ray, rays) per chunk - measurement tools read ray state from it
    (trace_iterations below).

    deferred (a list, with rounds_state): the call enqueues the guessed round prefix and returns WITHOUT reading the
    counters back; it appends a callable that, invoked once the work has completed, tells whether that prefix was the
    whole trace (and refreshes the guess).  For callers that trace ahead of time on a side stream
    (training/step.py:prefetch_trace) and cannot afford a host sync there."""
    lib = _lib.lib()
    n = origins.shape[0]
    dev = origins.device
    pts = torch.empty(n, 3, device=dev, dtype=torch.float32)
    hit = torch.empty(n, device=dev, dtype=torch.uint8)
    dist = torch.empty(n, device=dev, dtype=torch.float32)
    rounds = lib.nefii_trace_max_rounds(ctypes.byref(params))
    # (inside a stream capture nothing may be read back: a capture of the non-adaptive path runs without the audit - its counters
    # would need a host copy - and the caller audits an eager trace of the same weights instead)
    if audit is not None and rounds_state is None and torch.cuda.is_current_stream_capturing():
        audit = None
    need_cnt = want_counters or rounds_state is not None or audit is not None
    groups = max(1, min(int(groups), n // 64)) if n > 0 else 1
    counters = torch.zeros(groups, rounds, _lib.TRACE_COUNTERS, device=dev, dtype=torch.int32) if need_cnt else None
    if n > 0:
        om = object_mask.to(torch.uint8).contiguous()
        per = -(-n // groups)
        per = -(-per // 64) * 64
        bounds = [(g * per, min(n, (g + 1) * per)) for g in range(groups) if g * per < n]
        cur = torch.cuda.current_stream(dev)
        streams = [cur] if len(bounds) == 1 else _trace_streams(dev, len(bounds))
        if len(bounds) > 1:
            start = cur.record_event()
        work = []
        for g, (lo, hi) in enumerate(bounds):
            with torch.cuda.stream(streams[g]):
                if len(bounds) > 1:
                    streams[g].wait_event(start)
                nbytes = lib.nefii_trace_workspace_bytes(hi - lo, ctypes.byref(params))
                work.append((torch.empty(nbytes, device=dev, dtype=torch.uint8), nbytes))

        if keep_workspace is not None:
            keep_workspace.extend((w, lo, hi - lo) for (w, _), (lo, hi) in zip(work, bounds))
        G = len(bounds)
        begin = (ctypes.c_int64 * (G + 1))(*([lo for lo, _ in bounds] + [n]))
        ws_ptrs = (ctypes.c_void_p * G)(*[w.data_ptr() for w, _ in work])
        ws_bytes = (ctypes.c_size_t * G)(*[b for _, b in work])
        st_ptrs = (ctypes.c_void_p * G)(*[st.cuda_stream for st in streams])

        def run(sel, r0, r1):
            """rounds [r0, r1) of the chunks in `sel`, enqueued round-major by one library call"""
            if len(sel) == G and G > 1:
                b_, w_, wb_, s_, c_ = begin, ws_ptrs, ws_bytes, st_ptrs, counters
                _lib.check(lib.nefii_trace_rays_groups(
                    ctypes.byref(pm_sdf.struct), ctypes.byref(params), _ptr(origins), _ptr(dirs), _ptr(om), G, b_,
                    _ptr(lin_steps), _ptr(minsdf_steps), _ptr(pts), _ptr(hit), _ptr(dist), w_, wb_,
                    _ptr(c_) if c_ is not None else None, r0, r1, s_), 'nefii_trace_rays_groups')
                return
            for g in sel:
                lo, hi = bounds[g]
                ws, nbytes = work[g]
                _lib.check(lib.nefii_trace_rays_rounds(
                    ctypes.byref(pm_sdf.struct), ctypes.byref(params), _ptr(origins[lo:hi]), _ptr(dirs[lo:hi]),
                    _ptr(om[lo:hi]), hi - lo, _ptr(lin_steps), _ptr(minsdf_steps), _ptr(pts[lo:hi]), _ptr(hit[lo:hi]),
                    _ptr(dist[lo:hi]), _ptr(ws), nbytes, _ptr(counters[g]) if counters is not None else None,
                    r0, r1, streams[g].cuda_stream), 'nefii_trace_rays_rounds')

        def join():
            for st in streams:
                if st is not cur:
                    cur.wait_event(st.record_event())

        everyone = list(range(G))
        if rounds_state is None:
            run(everyone, 0, 0)
            join()
            if audit is not None:
                # the non-adaptive path has no counter read-back of its own.  A caller that hands in a `deferred` list gets the
                # audit there (pinned asynchronous copy, run once the work has completed); otherwise one host sync
                if deferred is not None:
                    ahead = torch.empty(counters.shape, dtype=counters.dtype, pin_memory=True)
                    ahead.copy_(counters, non_blocking=True)
                    deferred.append(lambda: audit(_audit_of(ahead), _lip_audit_of(ahead)))
                else:
                    (lambda h: audit(_audit_of(h), _lip_audit_of(h)))(counters.cpu())
        else:
            guess = rounds if rounds_state.guess is None else max(2, min(rounds, rounds_state.guess))
            run(everyone, 0, guess)
            join()
            if deferred is not None:
                # the counters travel to pinned host memory behind the enqueued rounds, on the stream of the trace: the
                # check then reads them without a copy on the CALLER'S stream (which would wait for whatever that stream
                # still has to run - the previous step's tail)
                ahead = None
                if os.environ.get('NEFII_COUNTERS_AHEAD', '1') != '0':
                    ahead = torch.empty(counters.shape, dtype=counters.dtype, pin_memory=True)
                    ahead.copy_(counters, non_blocking=True)

                def check():
                    """call once the enqueued prefix has completed: None if it was the whole trace, else the remaining
                    rounds are run (same streams, same workspaces) and the refreshed (points, hit, dists) returned"""
                    host = ahead if ahead is not None else counters.cpu()
                    again = [g for g in everyone if guess < rounds and int(host[g, guess - 1, _WORK].sum()) > 0]
                    if again:
                        run(again, guess, 0)
                        for st in streams:
                            st.synchronize()
                        host = counters.cpu()
                    busy = torch.nonzero(host[:, :, _WORK].sum(dim=(0, 2))).flatten()
                    rounds_state.guess = (int(busy[-1]) if busy.numel() else 0) + 3
                    if audit is not None:
                        audit(_audit_of(host), _lip_audit_of(host))
                    return (pts, hit.bool(), dist) if again else None
                deferred.append(check)
                if want_counters:
                    return pts, hit.bool(), dist, sum_counters(counters)
                return pts, hit.bool(), dist
            host = counters.cpu()                            # the one host sync (the caller syncs next anyway)
            if guess < rounds:
                again = [g for g in everyone if int(host[g, guess - 1, _WORK].sum()) > 0]
                if again:
                    run(again, guess, 0)
                    join()
                    host = counters.cpu()
            busy = torch.nonzero(host[:, :, _WORK].sum(dim=(0, 2))).flatten()
            last = int(busy[-1]) if busy.numel() else 0
            rounds_state.guess = last + 3                    # last emitting round + its consumer + one spare
            if audit is not None:
                audit(_audit_of(host), _lip_audit_of(host))
    if want_counters:
        return pts, hit.bool(), dist, sum_counters(counters)
    return pts, hit.bool(), dist


def trace_iterations(kept):
    """Sphere-tracing iterations each ray took, from the workspaces a finished trace_rays(keep_workspace=kept) left: the
    tracer parks the count in bits 20-23 of a ray's flag word when its sphere tracing ends (csrc/nefii_tracer.hip; the flag
    array follows 13 float arrays of n rays, each padded to 256 bytes).  Measurement only (tools/tier_parity.py)."""
    out = []
    for ws, _lo, n in kept:
        stride = (4 * n + 255) // 256 * 256
        flags = ws[13 * stride:13 * stride + 4 * n].view(torch.int32)
        out.append((flags >> 20) & 0xF)
    return torch.cat(out)


def camera_rays(uv, pose, intrinsics):
    """uv [B,S,2], pose [B,4,4], K [B,4,4] -> dirs [B,S,3], per-ray origins [B,S,3]."""
    lib = _lib.lib()
    B, S, _ = uv.shape
    dirs = torch.empty(B, S, 3, device=uv.device, dtype=torch.float32)
    orig = torch.empty(B, S, 3, device=uv.device, dtype=torch.float32)
    uv_c, pose_c, k_c = _f32(uv), _f32(pose), _f32(intrinsics)      # converted copies must outlive the launch call
    _lib.check(lib.nefii_camera_rays(_ptr(uv_c), _ptr(pose_c), _ptr(k_c), B, S, _ptr(dirs), _ptr(orig), _stream()),
               'nefii_camera_rays')
    return dirs, orig


class AssembleRowsFn(torch.autograd.Function):
    """Per-ray output buffers in two launches (include/nefii_amd.h: nefii_assemble_rows): output k is a [rows, cols[k]]
    buffer of fills[k] whose rows where[i] take row i of srcs[k] ([n, cols[k]], or [1, cols[k]] / [n, 1] broadcast).
    Backward: one launch gathers the rows of every output gradient that exists."""

    @staticmethod
    def forward(ctx, where, rows, fills, cols, *srcs):
        lib = _lib.lib()
        n = where.shape[0]
        dev = where.device
        blocks = (_lib.RowBlock * len(srcs))()
        outs, keep, shapes = [], [], []
        for k, src in enumerate(srcs):
            shapes.append(tuple(src.shape))
            s = _f32(src)
            if s.shape[-1] != cols[k]:          # a column broadcast ([n, 1] -> [n, C]) is materialised; a row broadcast is a stride
                s = s.expand(s.shape[0], cols[k]).contiguous()
            keep.append(s)
            out = torch.empty(rows, cols[k], device=dev, dtype=torch.float32)
            outs.append(out)
            blocks[k] = _lib.RowBlock(_ptr(s), _ptr(out), cols[k], 0 if s.shape[0] == 1 and n != 1 else cols[k], float(fills[k]), 0)
        _lib.check(lib.nefii_assemble_rows(blocks, len(srcs), _ptr(where), n, rows, _stream()), 'nefii_assemble_rows')
        ctx.save_for_backward(where)
        ctx.meta = (rows, tuple(cols), shapes)
        ctx.set_materialize_grads(False)        # a buffer the loss does not read sends None back, not zeros
        # a buffer assembled from constants (normals, points of frozen geometry) is a constant: without this every output of
        # the node would require grad because some other does
        ctx.mark_non_differentiable(*[o for k, o in enumerate(outs) if not ctx.needs_input_grad[4 + k]])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.lib()
        where, = ctx.saved_tensors
        rows, cols, shapes = ctx.meta
        n = where.shape[0]
        blocks = (_lib.RowBlock * len(grads))()
        res, keep = [], []
        for k, g in enumerate(grads):
            if g is None or not ctx.needs_input_grad[4 + k]:
                blocks[k] = _lib.RowBlock(None, None, cols[k], cols[k], 0.0, 0)
                res.append(None)
                continue
            gc = _f32(g)
            keep.append(gc)
            d = torch.empty(n, cols[k], device=where.device, dtype=torch.float32)
            blocks[k] = _lib.RowBlock(_ptr(gc), _ptr(d), cols[k], cols[k], 0.0, 0)
            res.append(d)
        if any(r is not None for r in res) and n > 0:
            _lib.check(lib.nefii_gather_rows(blocks, len(grads), _ptr(where), n, rows, _stream()), 'nefii_gather_rows')
        for k, r in enumerate(res):             # undo the broadcasts of forward
            if r is not None and tuple(r.shape) != shapes[k]:
                r = r.sum(dim=0, keepdim=True) if shapes[k][0] == 1 and n != 1 else r
                r = r.sum(dim=1, keepdim=True) if shapes[k][-1] == 1 and cols[k] != 1 else r
                res[k] = r.reshape(shapes[k])
        return (None, None, None, None) + tuple(res)


class MaterialHeadGlobalFn(torch.autograd.Function):
    """(roughness [1,1], specular [1,3]) from the GLOBAL roughness / specular parameters of physg.conf's material network
    (nefii_material_head_global): sigmoid, TINNY_ROUGHNESS blend, white-specular expand, warm-up flags and the 0.16 s^2 remap
    in one launch each way."""

    @staticmethod
    def forward(ctx, rough_param, spec_param, fake_rough, fake_spec):
        r_, s_ = _f32(rough_param), _f32(spec_param)
        rough = torch.empty(1, 1, device=r_.device, dtype=torch.float32)
        spec = torch.empty(1, 3, device=r_.device, dtype=torch.float32)
        _lib.check(_lib.lib().nefii_material_head_global(_ptr(r_), _ptr(s_), s_.numel(), int(bool(fake_rough)), int(bool(fake_spec)),
                                                         _ptr(rough), _ptr(spec), _stream()), 'nefii_material_head_global')
        ctx.save_for_backward(r_, s_)
        ctx.flags = (int(bool(fake_rough)), int(bool(fake_spec)))
        ctx.shapes = (rough_param.shape, spec_param.shape)
        ctx.set_materialize_grads(False)
        return rough, spec

    @staticmethod
    def backward(ctx, d_rough, d_spec):
        r_, s_ = ctx.saved_tensors
        if d_rough is None and d_spec is None:
            return None, None, None, None
        dr = _f32(d_rough) if d_rough is not None else None
        ds = _f32(d_spec) if d_spec is not None else None
        gr, gs = torch.empty_like(r_), torch.empty_like(s_)
        _lib.check(_lib.lib().nefii_material_head_global_backward(_ptr(r_), _ptr(s_), s_.numel(), ctx.flags[0], ctx.flags[1],
                                                                  _ptr(dr), _ptr(ds), _ptr(gr), _ptr(gs), _stream()),
                   'nefii_material_head_global_backward')
        return gr.reshape(ctx.shapes[0]), gs.reshape(ctx.shapes[1]), None, None


def prepare_hits(points, ray_dirs, grad, feat, idx):
    """(points[idx], view = -ray_dirs[idx] / (norm + 1e-6), normals = grad[idx] / (norm + 1e-6), feat[idx] or None) in one
    launch (nefii_prepare_hits); constants of the autograd graph (frozen geometry)."""
    n = idx.shape[0]
    dev = points.device
    pts = torch.empty(n, 3, device=dev, dtype=torch.float32)
    view = torch.empty(n, 3, device=dev, dtype=torch.float32)
    nrm = torch.empty(n, 3, device=dev, dtype=torch.float32)
    fcols = 0 if feat is None else feat.shape[1]
    fout = None if feat is None else torch.empty(n, fcols, device=dev, dtype=torch.float32)
    if n > 0:
        p_, d_, g_ = _f32(points), _f32(ray_dirs), _f32(grad)
        f_ = None if feat is None else _f32(feat)
        _lib.check(_lib.lib().nefii_prepare_hits(_ptr(p_), _ptr(d_), _ptr(g_), _ptr(f_), fcols, _ptr(idx), n, p_.shape[0],
                                                 _ptr(pts), _ptr(view), _ptr(nrm), _ptr(fout), _stream()), 'nefii_prepare_hits')
    return pts, view, nrm, fout


def assemble_rows(where, rows, fills, cols, srcs):
    return AssembleRowsFn.apply(where, int(rows), tuple(fills), tuple(cols), *srcs)


class SGRenderFn(torch.autograd.Function):
    """render_with_sg for one base material; differentiable wrt lgtSGs, specular, roughness, albedo."""

    @staticmethod
    def forward(ctx, lgt, spec, rough, albedo, normal, view):
        lib = _lib.lib()
        n = normal.shape[0]
        lgt_c, spec_c, rough_c = _f32(lgt), _f32(spec.expand(1, 3)), _f32(rough)
        albedo_c, normal_c, view_c = _f32(albedo), _f32(normal), _f32(view)
        rgb = torch.empty(n, 3, device=normal.device, dtype=torch.float32)
        srgb, drgb = torch.empty_like(rgb), torch.empty_like(rgb)
        _lib.check(lib.nefii_sg_render_forward(_ptr(lgt_c), lgt_c.shape[0], _ptr(spec_c), _ptr(rough_c), _ptr(albedo_c),
                                               _ptr(normal_c), _ptr(view_c), n, _ptr(rgb), _ptr(srgb), _ptr(drgb),
                                               _stream()), 'nefii_sg_render_forward')
        ctx.save_for_backward(lgt_c, spec_c, rough_c, albedo_c, normal_c, view_c)
        ctx.spec_shape = tuple(spec.shape)
        ctx.set_materialize_grads(False)        # the specular / diffuse parts usually take no part in the loss: NULL, not zeros
        return rgb, srgb, drgb

    @staticmethod
    def backward(ctx, d_rgb, d_s, d_d):
        lib = _lib.lib()
        lgt, spec, rough, albedo, normal, view = ctx.saved_tensors
        n = normal.shape[0]
        if d_rgb is None and d_s is None and d_d is None:
            return None, None, None, None, None, None
        g_alb = torch.empty_like(albedo)
        acc = torch.zeros(4 + lgt.numel(), device=albedo.device, dtype=torch.float32)      # the accumulators: one fill
        g_rough, g_spec, g_lgt = acc[0:1].view(1, 1), acc[1:4].view(1, 3), acc[4:].view_as(lgt)
        if d_rgb is None:
            d_rgb = torch.zeros(n, 3, device=albedo.device, dtype=torch.float32)
        d_rgb, d_s, d_d = _f32(d_rgb), _f32(d_s), _f32(d_d)     # keep converted copies alive across the launch call
        _lib.check(lib.nefii_sg_render_backward(_ptr(lgt), lgt.shape[0], _ptr(spec), _ptr(rough), _ptr(albedo),
                                                _ptr(normal), _ptr(view), n, _ptr(d_rgb), _ptr(d_s),
                                                _ptr(d_d), _ptr(g_alb), _ptr(g_rough), _ptr(g_spec), _ptr(g_lgt),
                                                _stream()), 'nefii_sg_render_backward')
        if ctx.spec_shape[-1] == 1:
            g_spec = g_spec.sum(-1, keepdim=True)
        return g_lgt, g_spec, g_rough, g_alb, None, None


class EnvRadianceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lgt, dirs, eps):
        lib = _lib.lib()
        lgt_c, dirs_c = _f32(lgt), _f32(dirs)
        n = dirs_c.shape[0]
        rgb = torch.empty(n, 3, device=dirs.device, dtype=torch.float32)
        _lib.check(lib.nefii_env_radiance_forward(_ptr(lgt_c), lgt_c.shape[0], _ptr(dirs_c), n, eps, _ptr(rgb),
                                                  _stream()), 'nefii_env_radiance_forward')
        ctx.save_for_backward(lgt_c, dirs_c)
        ctx.eps = eps
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        lib = _lib.lib()
        lgt, dirs = ctx.saved_tensors
        g = torch.zeros_like(lgt)
        d_rgb = _f32(d_rgb)
        _lib.check(lib.nefii_env_radiance_backward(_ptr(lgt), lgt.shape[0], _ptr(dirs), dirs.shape[0], ctx.eps,
                                                   _ptr(d_rgb), _ptr(g), _stream()), 'nefii_env_radiance_backward')
        return g, None, None


# ---- descriptors of the reference's three networks ------------------------------------------------
def sdf_specs(cfg, feature_vector_size):
    """LayerSpec list for ImplicitNetwork (implicit_differentiable_renderer.py:18-83)."""
    dims = list(cfg['dims'])
    L = int(cfg.get('multires', 0))
    d0 = 3 + 6 * L
    if cfg.get('use_last_as_f', False):
        full = [d0] + dims + [cfg['d_out']]
    else:
        full = [d0] + dims + [cfg['d_out'] + feature_vector_size]
    skip = tuple(cfg.get('skip_in', ()))
    specs = []
    prev_out = None
    for l in range(len(full) - 1):
        n_out = full[l + 1] - d0 if (l + 1) in skip else full[l + 1]
        if l == 0:
            specs.append(LayerSpec(n_out, d0, e_src0=0, e_len=d0))
        elif l in skip:
            specs.append(LayerSpec(n_out, prev_out + d0, x_src0=0, x_len=prev_out, e_src0=prev_out, e_len=d0,
                                   scale=1.0 / math.sqrt(2)))
        else:
            specs.append(LayerSpec(n_out, prev_out, x_src0=0, x_len=prev_out))
        prev_out = n_out
    return specs, [L, -1, -1]


def radiance_specs(cfg, feature_vector_size):
    """RenderingNetwork (implicit_differentiable_renderer.py:126-241): layer 0 eats cat[PE(x), PE(v), n, feat]."""
    mode = cfg.get('mode', 'idr')
    Lv, Lx = int(cfg.get('multires_view', 0)), int(cfg.get('multires_xyz', 0))
    if mode == 'idr':
        enc = [Lx, Lv, 0]
    elif mode == 'no_view_dir':
        enc = [Lx, 0, -1]
    elif mode == 'no_normal':
        enc = [Lx, Lv, -1]
    else:
        raise ValueError(mode)
    ew = sum(3 + 6 * e for e in enc if e >= 0)
    dims = [ew + feature_vector_size] + list(cfg['dims']) + [cfg['d_out']]
    specs = [LayerSpec(dims[1], dims[0], x_src0=ew, x_len=feature_vector_size, e_src0=0, e_len=ew)]
    for l in range(1, len(dims) - 1):
        specs.append(LayerSpec(dims[l + 1], dims[l], x_src0=0, x_len=dims[l]))
    if cfg.get('normalize_output', True):
        head = HEAD_TANH01
    elif not cfg.get('clip_output', False):
        head = HEAD_NONE
    else:
        head = {'relu': HEAD_RELU, 'abs': HEAD_ABS, 'relu_init': HEAD_RELU_INIT, 'pow2': HEAD_POW2}[
            cfg.get('clip_method', 'relu')]
    return specs, enc, head


def material_specs(cfg, feature_vector_size, dim_out):
    """EnvmapMaterialNetwork.diffuse_albedo_layers (sg_envmap_material.py:92-103): cat[PE(x), feat] -> ELU MLP."""
    Lx = int(cfg.get('multires', 0))
    ew = 3 + 6 * Lx
    dims = [ew + feature_vector_size] + list(cfg['dims']) + [dim_out]
    specs = [LayerSpec(dims[1], dims[0], x_src0=ew, x_len=feature_vector_size, e_src0=0, e_len=ew)]
    for l in range(1, len(dims) - 1):
        specs.append(LayerSpec(dims[l + 1], dims[l], x_src0=0, x_len=dims[l]))
    return specs, [Lx, -1, -1]


# ---- Monte-Carlo direct + indirect shading ----------------------------------------------------------
def mis_sample(lgt, rough, normal, view, uniforms):
    """-> wi [3,n,3], own_pdf [3,n], pdf_table [3,n,3]   (no gradient: the reference samples under no_grad)."""
    lib = _lib.lib()
    n = normal.shape[0]
    dev = normal.device
    wi = torch.empty(3, n, 3, device=dev, dtype=torch.float32)
    own = torch.empty(3, n, device=dev, dtype=torch.float32)
    tab = torch.empty(3, n, 3, device=dev, dtype=torch.float32)
    lgt_c = _f32(lgt)
    rough_c, normal_c, view_c, uni_c = _f32(rough).reshape(-1), _f32(normal), _f32(view), _f32(uniforms)
    _lib.check(lib.nefii_mis_sample(_ptr(lgt_c), lgt_c.shape[0], _ptr(rough_c), _ptr(normal_c), _ptr(view_c),
                                    _ptr(uni_c), n, _ptr(wi), _ptr(own), _ptr(tab), _stream()), 'nefii_mis_sample')
    return wi, own, tab


class McShadeFn(torch.autograd.Function):
    """Sum over the 3 MIS samples of (direct*vis + (1-vis)*indirect) x (GGX specular + Lambert);
    differentiable wrt light, indirect, albedo, roughness and (if it requires grad) the global specular."""

    @staticmethod
    def forward(ctx, spec, rough, albedo, normal, view, wi, own, tab, light, vis, indirect):
        lib = _lib.lib()
        n = normal.shape[0]
        t = [_f32(spec.expand(1, 3)), _f32(rough).reshape(-1), _f32(albedo), _f32(normal), _f32(view), _f32(wi),
             _f32(own), _f32(tab), _f32(light), _f32(vis), _f32(indirect)]
        rgb = torch.empty(n, 3, device=normal.device, dtype=torch.float32)
        srgb, drgb = torch.empty_like(rgb), torch.empty_like(rgb)
        _lib.check(lib.nefii_mc_shade_forward(*[_ptr(x) for x in t], n, _ptr(rgb), _ptr(srgb), _ptr(drgb), _stream()),
                   'nefii_mc_shade_forward')
        ctx.save_for_backward(*t)
        ctx.spec_grad = spec.requires_grad
        ctx.spec_shape = tuple(spec.shape)
        ctx.rough_shape = tuple(rough.shape)
        return rgb, srgb, drgb

    @staticmethod
    def backward(ctx, d_rgb, d_s, d_d):
        lib = _lib.lib()
        t = ctx.saved_tensors
        n = t[3].shape[0]
        dev = t[3].device
        g_light = torch.empty(3, n, 3, device=dev, dtype=torch.float32)
        g_ind = torch.empty_like(g_light)
        g_alb = torch.empty(n, 3, device=dev, dtype=torch.float32)
        g_rough = torch.empty(n, device=dev, dtype=torch.float32)
        g_spec = torch.zeros(1, 3, device=dev, dtype=torch.float32) if ctx.spec_grad else None
        d_rgb, d_s, d_d = _f32(d_rgb), _f32(d_s), _f32(d_d)
        _lib.check(lib.nefii_mc_shade_backward(*[_ptr(x) for x in t], n, _ptr(d_rgb), _ptr(d_s),
                                               _ptr(d_d), _ptr(g_light), _ptr(g_ind), _ptr(g_alb), _ptr(g_rough),
                                               _ptr(g_spec), _stream()), 'nefii_mc_shade_backward')
        if g_spec is not None and ctx.spec_shape[-1] == 1:
            g_spec = g_spec.sum(-1, keepdim=True)
        return (g_spec, g_rough.reshape(ctx.rough_shape), g_alb, None, None, None, None, None, g_light, None, g_ind)


class IdrLossFn(torch.autograd.Function):
    """IDRLoss value + gradient in one launch (nefii_idr_loss).  Returns losses [6] = (loss, idr_rgb_loss, sg_rgb_loss,
    mask_loss, normalsmooth_loss, background_rgb_loss); only losses[0] is differentiable, wrt idr_rgb and sg_rgb."""

    @staticmethod
    def forward(ctx, idr_rgb, sg_rgb, rgb_gt, net_mask, obj_mask, sdf_output, normals, params):
        lib = _lib.lib()
        n = sg_rgb.shape[0]
        dev = sg_rgb.device
        losses = torch.empty(6, device=dev, dtype=torch.float32)
        need_i, need_s = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_idr = torch.empty(n, 3, device=dev, dtype=torch.float32) if need_i else None
        d_sg = torch.empty(n, 3, device=dev, dtype=torch.float32) if need_s else None
        # converted copies are bound to names: a temporary handed to _ptr() dies before the launch call is made and the
        # next temporary of the same size is given its memory
        idr_c, sg_c, gt_c = _f32(idr_rgb), _f32(sg_rgb), _f32(rgb_gt).reshape(-1, 3)
        net_c, obj_c = net_mask.to(torch.uint8).contiguous(), obj_mask.to(torch.uint8).contiguous()
        sdf_c = _f32(sdf_output).reshape(-1)
        nrm_c = _f32(normals) if normals is not None else None
        _lib.check(lib.nefii_idr_loss(ctypes.byref(params), _ptr(idr_c), _ptr(sg_c), _ptr(gt_c), _ptr(net_c), _ptr(obj_c),
                                      _ptr(sdf_c), _ptr(nrm_c), n, _ptr(losses), _ptr(d_idr), _ptr(d_sg), _stream()),
                   'nefii_idr_loss')
        ctx.save_for_backward(d_idr, d_sg)
        return losses

    @staticmethod
    def backward(ctx, g):
        d_idr, d_sg = ctx.saved_tensors
        s = g[0]
        return (d_idr * s if d_idr is not None else None, d_sg * s if d_sg is not None else None,
                None, None, None, None, None, None)
