"""nefii_amd - MI355X-native (gfx950) implementation of NeFII's per-ray-batch inverse-rendering hot path.

Layout: csrc/ (HIP kernels + the C ABI of include/nefii_amd.h), _lib.py (ctypes binding), ops.py (tensor
wrappers + autograd glue), model/ + utils/ (host-side mirror of the reference's Python interface),
conf.py (HOCON subset), synthetic.py (procedural workloads for BASELINE.json's configs).
"""
__version__ = '0.1.0'
