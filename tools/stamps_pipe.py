"""Per-layer timeline of the PIPELINED single-pass tile evaluator (mlp_tile.h "16s3") from s_memtime stamps, in shader clock
cycles: build a stamped library (hipcc ... -DNEFII_STAMPS), point NEFII_LIB_PATH at it.  Stamps per layer (1 ..) and wave:
0 layer start, 1 block A done (K1 beside the previous layer's group-b epilogue), 2 released by the barrier, 3 block B1 done,
4 block B2 done (beside this layer's group-a epilogue)."""
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
h = ctypes.CDLL(_lib.LIB_PATH)
h.nefii_debug_stamps.argtypes = [ctypes.c_void_p]
buf = np.zeros(2 * 8 * 12 * 5, dtype=np.uint64)
for it in range(3):
    h.nefii_debug_stamps(buf.ctypes.data)
    ops.sdf_eval(pm, x, coarse=True); torch.cuda.synchronize()
h.nefii_debug_stamps(buf.ctypes.data)
t = buf.reshape(2, 8, 12, 5).astype(np.int64)
NL = len(specs) - 1
for tile in range(2):
    print('tile', tile, '(cycles): per wave 0..7')
    for l in range(1, NL):
        T = t[tile, :, l, :]
        nxt = t[tile, :, l + 1, 0] if l + 1 < NL else None
        print('L%d  block A %s\n    barrier 1 %s\n    block B1 %s\n    block B2 %s%s' % (
            l, (T[:, 1] - T[:, 0]).tolist(), (T[:, 2] - T[:, 1]).tolist(), (T[:, 3] - T[:, 2]).tolist(),
            (T[:, 4] - T[:, 3]).tolist(), '' if nxt is None else '\n    to the next layer start (barrier 2 + layer top) %s   layer period %s' % (
                (nxt - T[:, 4]).tolist(), (nxt - T[:, 0]).tolist())))
