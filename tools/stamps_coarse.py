"""Per-layer timeline of the single-pass (coarse) tile evaluator (mlp_tile.h "16s") from s_memtime stamps, in shader
clock cycles.  Build a stamped library (hipcc ... -DNEFII_STAMPS) and point NEFII_LIB_PATH at it; NEFII_COARSE_QT=4|6|8.
Stamps per layer and wave: 0 layer start, 1 k-loop done, 2 epilogue (and, with two activation images: stores) done = at
the barrier, 3 released, 4 outputs stored + second barrier (the big single-image tiles of NEFII_COARSE_QT=6|8 only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from nefii_amd import ops, synthetic as syn, _lib
from oracle import nets
mc = syn.model_conf(os.environ.get('MODEL', 'physg'))
sd = syn.make_state_dict(mc, seed=0, bumpy=0.004)
specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
x = (torch.randn(12 * 256 * 64, 3) * 0.4).cuda()
lib = _lib.lib()
h = ctypes.CDLL(_lib.LIB_PATH)
h.nefii_debug_stamps.argtypes = [ctypes.c_void_p]
buf = np.zeros(2 * 8 * 12 * 5, dtype=np.uint64)
for it in range(3):
    h.nefii_debug_stamps(buf.ctypes.data)      # reset the tile counter
    ops.sdf_eval(pm, x, coarse=True); torch.cuda.synchronize()
h.nefii_debug_stamps(buf.ctypes.data)
t = buf.reshape(2, 8, 12, 5).astype(np.int64)
NL = len(specs) - 1
for tile in range(2):
    t0 = t[tile, :, 0, 0].min()
    print('tile', tile, '(cycles): per wave 0..7')
    for l in range(NL):
        T = t[tile, :, l, :] - t0
        print('L%d start %s\n   k-loop   %s\n   epilogue (+ stores, two images) %s\n   barrier  %s\n   stores + second barrier (single image) %s' % (
            l, T[:, 0].tolist(), (T[:, 1] - T[:, 0]).tolist(), (T[:, 2] - T[:, 1]).tolist(), (T[:, 3] - T[:, 2]).tolist(),
            (T[:, 4] - T[:, 3]).tolist()))
    print('tile span', t[tile, :, NL - 1, 4].max() - t0)
