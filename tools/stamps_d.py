"""Timeline of the four-wave single-pass tile ("16d": two independent groups per workgroup; "wg" below = 2 x workgroup + group)
from s_memtime stamps (shader-clock cycles): for
the two groups of a CU, the phases of each layer of their third tile side by side.
Build: tools/ab_build.sh stamps -DNEFII_STAMPS; run: NEFII_COARSE_D=1 NEFII_LIB_PATH=build_ab/libnefii_stamps.so python tools/stamps_d.py
Stamps per layer (wave 0): 0 layer start, 1 k-loop done, 2 past the first barrier, 3 epilogue + stores done, 4 past the second."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from nefii_amd import ops, synthetic as syn, _lib
from oracle import nets
mc = syn.model_conf('physg')
sd = syn.make_state_dict(mc, seed=0, bumpy=0.0, scene=os.environ.get('SCENE', 'bowl_trained'))
specs, enc = ops.sdf_specs(mc['implicit_network'], mc['feature_vector_size'])
pm = ops.PackedMLP(specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, enc, 0, 'cuda', f16x3=True)
ws, bs = zip(*[nets.linear_params(sd, 'implicit_network.lin%d' % l) for l in range(len(specs))])
pm.pack([w.cuda() for w in ws], [b.cuda() for b in bs])
x = (torch.randn(12 * 256 * 64, 3) * 0.4).cuda()
_lib.lib()
h = ctypes.CDLL(_lib.LIB_PATH)
h.nefii_debug_dstamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
for it in range(3):
    ops.sdf_eval(pm, x, coarse=True); torch.cuda.synchronize()
buf = np.zeros(512 * 12 * 5, dtype=np.uint64)
hw = np.zeros(1024, dtype=np.uint32)
h.nefii_debug_dstamps(buf.ctypes.data, hw.ctypes.data)
t = buf.reshape(512, 12, 5).astype(np.int64)
hw = hw.reshape(512, 2)
NL = len(specs) - 1
# HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13 (gfx9); XCC_ID bits 3:0
key = [(int(hw[b, 1] & 15), int((hw[b, 0] >> 13) & 7), int((hw[b, 0] >> 12) & 1), int((hw[b, 0] >> 8) & 15)) for b in range(512)]
groups = {}
for b, k in enumerate(key):
    groups.setdefault(k, []).append(b)
sizes = sorted(len(v) for v in groups.values())
print('workgroups per (xcc, se, sh, cu): %d CUs, min %d max %d' % (len(groups), sizes[0], sizes[-1]))
shown = 0
for k, blocks in sorted(groups.items()):
    if len(blocks) != 2 or shown >= 3:
        continue
    shown += 1
    a, b = blocks
    t0 = min(t[a, 0, 0], t[b, 0, 0])
    print('CU %s: workgroups %d and %d (cycles from the earlier tile start)' % (k, a, b))
    for l in range(NL):
        for w in (a, b):
            T = t[w, l] - t0
            print('  L%d wg %3d  start %7d | k-loop %6d | barrier %5d | epilogue %6d | barrier %5d | -> %7d' % (
                l, w, T[0], T[1] - T[0], T[2] - T[1], T[3] - T[2], T[4] - T[3], T[4]))
# all workgroups: mean phase lengths of the 512-wide layers
mid = t[:, 1:NL, :]
print('mean over all workgroups, 512-wide layers: k-loop %.0f, first barrier %.0f, epilogue + stores %.0f, second barrier %.0f, layer period %.0f cycles' % (
    (mid[:, :, 1] - mid[:, :, 0]).mean(), (mid[:, :, 2] - mid[:, :, 1]).mean(), (mid[:, :, 3] - mid[:, :, 2]).mean(),
    (mid[:, :, 4] - mid[:, :, 3]).mean(), (t[:, NL - 1, 4] - t[:, 1, 0]).mean() / (NL - 1)))
