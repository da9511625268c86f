"""Import shim for the *reference* implementation (survey container only).

Used ONLY by tests/golden/make_golden.py to generate golden vectors from
/root/reference.  Nothing in the -m gpu tests, smoke() or bench.py imports this:
the reference never travels to the GPU box.  Recipe follows SURVEY.md Appendix B.
"""
import sys
import types

import torch

REF_ROOT = '/root/reference/code'


def install():
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    io = stub('imageio')
    pl = stub('imageio.plugins')
    fi = stub('imageio.plugins.freeimage', download=lambda: None)
    io.plugins = pl
    pl.freeimage = fi
    stub('skimage')
    stub('cv2')
    stub('kornia')
    # the reference hard-codes .cuda(); torch here is CPU only
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


class Conf(dict):
    """dict-backed stand-in for pyhocon.ConfigTree (pyhocon is not installed)."""

    def _g(self, k, d=None):
        cur = self
        for p in k.split('.'):
            if p not in cur:
                return d
            cur = cur[p]
        return cur

    def get_int(self, k, default=None):
        return int(self._g(k, default))

    def get_float(self, k, default=None):
        return float(self._g(k, default))

    def get_bool(self, k, default=None):
        return bool(self._g(k, default))

    def get_string(self, k, default=None):
        return str(self._g(k, default))

    def get_config(self, k):
        return Conf(self._g(k))
