"""CPU oracle for the NeFII per-ray-batch inverse-rendering hot path.

TEST INFRASTRUCTURE ONLY.  This package is a from-scratch fp32 PyTorch-CPU
restatement of the reference algorithm (FuxiComputerVision/Nefii, files cited
per function).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it - and there only as the
checker / the reported CPU baseline, never as the thing shipped.  The product
(``nefii_amd``) never imports it and fails loudly when its HIP library is
missing.

Pinning: the reference ships no tests or golden vectors, so
the oracle is pinned against outputs of the reference itself, generated in the
build container by ``tests/golden/make_golden.py`` (which imports
/root/reference through ``tests/golden/ref_shim.py``) and committed as small
``.npz`` fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every fixture.

Semantics that deliberately differ from the reference's batch behaviour:
  * bisection (``rootfind``) stops per ray, not when the whole batch has
    converged (reference ray_tracing.py:264-277 keeps updating every ray while
    any ray works) - differences are <= 1e-6 in t;
  * the tracer always returns ``points = origin + dist * dir``.
"""
