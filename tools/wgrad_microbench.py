"""nefii_mlp_wgrad_f16 (dW = dz^T x on fp16 MFMA, fp32 in / out) per call.  NEFII_WGRAD_TR=0 selects the scalar-load kernel.
Usage: python tools/wgrad_microbench.py [n ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefii_amd import _lib
from nefii_amd.ops import _ptr, _stream

dev = 'cuda'
for n in [int(a) for a in sys.argv[1:]] or [4096, 139264, 278528]:
    for n_out, k_in in ((512, 512), (512, 605)):
        dz = torch.randn(n, 512, device=dev) * 1e-6
        x = torch.randn(n, k_in, device=dev)
        S = torch.tensor([2.0 ** 26], device=dev)
        dW = torch.empty(n_out, k_in, device=dev)
        db = torch.empty(n_out, device=dev)
        call = lambda: _lib.check(_lib.lib().nefii_mlp_wgrad_f16(_ptr(dz), 512, _ptr(x), k_in, n, n_out, k_in, 1.0, _ptr(S),
                                                                 _ptr(dW), _ptr(db), _stream()), 'wgrad')
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('n %7d  %d x %d  %.3f ms per call  %.0f TFLOP/s  %.2f TB/s of fp32 operands' % (
            n, n_out, k_in, ms, 2.0 * n * n_out * k_in / ms / 1e9, n * (512 + k_in) * 4 / ms / 1e9))

# the same GEMMs fed with halves (nefii_mlp_wgrad_f16h: S dz in fp16; x = 16 h in fp16 for hidden layers, fp32 for layer 0)
for n in [int(a) for a in sys.argv[1:]] or [4096, 139264, 278528]:
    for n_out, k_in, x_half in ((512, 512, 1), (512, 605, 0)):
        S = torch.tensor([2.0 ** 26], device=dev)
        dz = (torch.randn(n, 512, device=dev) * 1e-6 * S).half()
        x = (torch.randn(n, k_in, device=dev) * 16).half() if x_half else torch.randn(n, k_in, device=dev)
        dW = torch.empty(n_out, k_in, device=dev)
        db = torch.empty(n_out, device=dev)
        call = lambda: _lib.check(_lib.lib().nefii_mlp_wgrad_f16h(_ptr(dz), 512, _ptr(x), k_in, x_half, n, n_out, k_in, 1.0,
                                                                  _ptr(S), _ptr(dW), _ptr(db), _stream()), 'wgrad_h')
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('halves: n %7d  %d x %d (x %s)  %.3f ms per call  %.0f TFLOP/s  %.2f TB/s of operands' % (
            n, n_out, k_in, 'fp16' if x_half else 'fp32', ms, 2.0 * n * n_out * k_in / ms / 1e9,
            n * (512 * 2 + k_in * (2 if x_half else 4)) / ms / 1e9))
