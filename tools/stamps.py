"""Per-layer timeline of the pipelined tile evaluator from s_memtime stamps (build with -DNEFII_STAMPS).
The stamps are compiled into the 32x32x16 instance (mlp_tile.h "16p"), so the script selects stream layout 0."""
import ctypes, os, sys
os.environ['NEFII_STREAM_LAYOUT'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from nefii_amd import ops, synthetic as syn, _lib
from oracle import nets
mc = syn.model_conf('physg')
NL = int(os.environ.get('NL', '8'))
if NL != 8:
    mc['implicit_network']['dims'] = [512] * NL
    mc['implicit_network']['skip_in'] = [NL // 2]
sd = syn.make_state_dict(mc, seed=0, bumpy=0.004)
specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
x = (torch.randn(8 * 256 * 64, 3) * 0.4).cuda()
lib = _lib.lib()

h = ctypes.CDLL(_lib.LIB_PATH)
h.nefii_debug_stamps.argtypes = [ctypes.c_void_p]
buf = np.zeros(2 * 8 * 12 * 5, dtype=np.uint64)
for it in range(3):
    h.nefii_debug_stamps(buf.ctypes.data)      # reset the tile counter
    ops.sdf_eval(pm, x); torch.cuda.synchronize()
h.nefii_debug_stamps(buf.ctypes.data)
t = buf.reshape(2, 8, 12, 5).astype(np.int64)
t0 = t[0, :, 0, 0].min()
for tile in range(2):
    print('tile', tile, ' (s_memtime ticks = 100 MHz? shown raw); rows = layers, cols: gemm | barrier1 | epilogue | barrier2, per wave 0..7')
    for l in range(min(NL, 8)):
        g = t[tile, :, l, 1] - t[tile, :, l, 0]; b1 = t[tile, :, l, 2] - t[tile, :, l, 1]
        e = t[tile, :, l, 3] - t[tile, :, l, 2]; b2 = t[tile, :, l, 4] - t[tile, :, l, 3]
        print('L%d start %7d | gemm %s | bar %s | epi %s | bar %s' % (l, t[tile, 0, l, 0] - t0, g.tolist(), b1.tolist(), e.tolist(), b2.tolist()))
    print('tile span', t[tile, :, NL - 1, 4].max() - t[tile, :, 0, 0].min())
