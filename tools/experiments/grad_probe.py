"""Which parameter gradient of config 3's shrunk parity test is the worst, and does the MLPs' fp16 training state change it?
Runs tests/test_gpu_configs.py::test_config_shrunk_in_pixels_vs_oracle[cfg3] with rel_l2 wrapped: every value is recorded,
values above the old 3e-3 bound are let through so that the test reaches its end.  NEFII_MLP_H16=0|1 selects the state.
(Result, end of round 4: 4.0e-3 on rendering_network.lin0.bias with either state - the replicated embedding, not the halves.)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import tests.test_gpu_configs as t
orig = t.rel_l2
vals = []
def rl(a, b):
    v = orig(a, b); vals.append(v); return min(v, 1e-9) if len(getattr(a, 'shape', ())) <= 2 and v < 1.0 and v > 2.5e-3 else v
t.rel_l2 = rl
try:
    t.test_config_shrunk_in_pixels_vs_oracle('cfg3')
except AssertionError as e:
    print('assert', str(e)[:200])
print('H16', os.environ.get('NEFII_MLP_H16', '1'), 'largest rel_l2 values seen', sorted(vals)[-6:])
