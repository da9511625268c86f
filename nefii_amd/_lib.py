"""ctypes binding of libnefii_hip.so (the C ABI declared in include/nefii_amd.h).

The product path has NO fallback: if the shared library is missing or cannot be loaded this
module raises, and every op built on it fails loudly.  Build with ``python -m nefii_amd.build``
(or ``__graft_entry__.build()``); the .so is kept in-tree next to the sources.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NEFII_LIB_PATH', os.path.join(HERE, 'csrc', 'libnefii_hip.so'))    # override: A/B builds

MAX_LAYERS = 12
TILE_ROWS = 32
MAX_WIDTH = 512
MAX_ENC = 96
ABI_VERSION = 15
TRACE_COUNTERS = 14         # int32 counters per tracer round (NEFII_TRACE_COUNTERS)

ACT_RELU, ACT_ELU, ACT_SOFTPLUS100 = 0, 1, 2
HEAD_NONE, HEAD_TANH01, HEAD_POW2, HEAD_SIGMOID, HEAD_RELU, HEAD_ABS, HEAD_RELU_INIT = range(7)

c_float_p = ctypes.c_void_p     # device pointers travel as integers


class Layer(ctypes.Structure):
    _fields_ = [('k_x', ctypes.c_int32), ('k_e', ctypes.c_int32), ('n_out', ctypes.c_int32),
                ('n_pad', ctypes.c_int32), ('w_fwd', ctypes.c_void_p), ('w_bwd', ctypes.c_void_p),
                ('bias', ctypes.c_void_p), ('w_f16x3', ctypes.c_void_p), ('w_bwd_f16x3', ctypes.c_void_p)]


class Mlp(ctypes.Structure):
    _fields_ = [('n_layers', ctypes.c_int32), ('act', ctypes.c_int32), ('head', ctypes.c_int32),
                ('enc_freqs', ctypes.c_int32 * 3), ('feat_width', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('w_stream', ctypes.c_void_p), ('layer', Layer * MAX_LAYERS)]


class TracerParams(ctypes.Structure):
    _fields_ = [('object_bounding_sphere', ctypes.c_float), ('sdf_threshold', ctypes.c_float),
                ('line_search_step', ctypes.c_float), ('line_step_iters', ctypes.c_int32),
                ('sphere_tracing_iters', ctypes.c_int32), ('n_steps', ctypes.c_int32),
                ('n_rootfind_steps', ctypes.c_int32), ('training', ctypes.c_int32), ('bisect_levels', ctypes.c_int32),
                ('precision', ctypes.c_int32), ('coarse_tau', ctypes.c_float), ('coarse_cap', ctypes.c_int32),
                ('minsdf_group', ctypes.c_int32), ('small_round', ctypes.c_int32), ('trace_tier', ctypes.c_int32),
                ('tier_kappa', ctypes.c_float), ('tier_gate', ctypes.c_float), ('minsdf_lipschitz', ctypes.c_float),
                ('unread_misses', ctypes.c_int32), ('split_fp8', ctypes.c_int32)]


class PackSource(ctypes.Structure):
    _fields_ = [('W', ctypes.c_void_p), ('bias', ctypes.c_void_p), ('n_out', ctypes.c_int32), ('k_in', ctypes.c_int32),
                ('x_src0', ctypes.c_int32), ('x_len', ctypes.c_int32), ('e_src0', ctypes.c_int32), ('e_len', ctypes.c_int32),
                ('scale', ctypes.c_float), ('skip_f32', ctypes.c_int32)]


class WgradItem(ctypes.Structure):
    _fields_ = [('dz16', ctypes.c_void_p), ('x', ctypes.c_void_p), ('dW', ctypes.c_void_p), ('db', ctypes.c_void_p),
                ('dz_stride', ctypes.c_int32), ('x_stride', ctypes.c_int32), ('x_half', ctypes.c_int32),
                ('n_out', ctypes.c_int32), ('k_in', ctypes.c_int32), ('scale', ctypes.c_float)]


MAX_WGRAD_ITEMS = 12


class RowBlock(ctypes.Structure):
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('cols', ctypes.c_int32), ('src_row_stride', ctypes.c_int32),
                ('fill', ctypes.c_float), ('reserved', ctypes.c_int32)]


MAX_ROW_BLOCKS = 12


class LossParams(ctypes.Structure):
    _fields_ = [('idr_rgb_weight', ctypes.c_float), ('sg_rgb_weight', ctypes.c_float), ('mask_weight', ctypes.c_float),
                ('alpha', ctypes.c_float), ('normalsmooth_weight', ctypes.c_float),
                ('background_rgb_weight', ctypes.c_float), ('loss_type', ctypes.c_int32),
                ('env_loss_type', ctypes.c_int32), ('r_patch', ctypes.c_int32), ('reserved', ctypes.c_int32)]


P = ctypes.c_void_p
I = ctypes.c_int
I64 = ctypes.c_int64
F = ctypes.c_float

# name -> (restype, argtypes); every symbol include/nefii_amd.h declares
SIGNATURES = {
    'nefii_abi_version': (I, []),
    'nefii_padded_width': (I, [I]),
    'nefii_pack_linear': (I, [P, P, I, I, I, I, I, I, F, P, P, P, P]),
    'nefii_pack_linear_f16x3': (I, [P, I, I, I, I, I, I, F, P, P]),
    'nefii_pack_linear_f16x3_bwd': (I, [P, I, I, I, I, I, I, F, P, P]),
    'nefii_pack_mlp': (I, [ctypes.POINTER(Mlp), ctypes.POINTER(PackSource), P]),
    'nefii_mlp_forward': (I, [ctypes.POINTER(Mlp), P, P, P, P, I64, P, I, P, I, P, I, P]),
    'nefii_mlp_backward': (I, [ctypes.POINTER(Mlp), P, I, P, I, I64, P, I, P]),
    'nefii_mlp_wgrad': (I, [P, I, P, I, I64, I, I, F, P, P, P]),
    'nefii_mlp_forward_f16': (I, [ctypes.POINTER(Mlp), P, P, P, P, I64, P, I, P, I, P, I, I, P]),
    'nefii_mlp_grad_scale': (I, [P, I64, P, P]),
    'nefii_mlp_backward_f16': (I, [ctypes.POINTER(Mlp), P, I, P, I, I64, P, I, P, P]),
    'nefii_mlp_wgrad_f16': (I, [P, I, P, I, I64, I, I, F, P, P, P, P]),
    'nefii_mlp_h16_supported': (I, [ctypes.POINTER(Mlp)]),
    'nefii_mlp_x0_width': (I, [ctypes.POINTER(Mlp)]),
    'nefii_mlp_forward_f16h': (I, [ctypes.POINTER(Mlp), P, P, P, P, I64, P, I, P, I, P, I, P, P, P]),
    'nefii_mlp_backward_f16h': (I, [ctypes.POINTER(Mlp), P, I, P, I, P, I64, P, I, P, P]),
    'nefii_mlp_wgrad_f16h': (I, [P, I, P, I, I, I64, I, I, F, P, P, P, P]),
    'nefii_mlp_wgrad_f16h_batch': (I, [ctypes.POINTER(WgradItem), I, I64, P, P]),
    'nefii_encode_inputs': (I, [ctypes.POINTER(Mlp), P, P, P, P, I64, P, I, P]),
    'nefii_sdf_value_grad': (I, [ctypes.POINTER(Mlp), P, I64, P, I, P, I, P, P, P]),
    'nefii_sdf_value_grad_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(Mlp), I64]),
    'nefii_sdf_stream_bytes': (ctypes.c_size_t, [ctypes.POINTER(Mlp)]),
    'nefii_pack_sdf_stream': (I, [ctypes.POINTER(Mlp), P, P]),
    'nefii_mlp_stream_bytes': (ctypes.c_size_t, [ctypes.POINTER(Mlp)]),
    'nefii_pack_mlp_stream': (I, [ctypes.POINTER(Mlp), P, P]),
    'nefii_sdf_eval': (I, [ctypes.POINTER(Mlp), P, I64, P, P]),
    'nefii_sdf_eval_coarse': (I, [ctypes.POINTER(Mlp), P, I64, P, P]),
    'nefii_sdf_coarse_supported': (I, [ctypes.POINTER(Mlp)]),
    'nefii_sdf_eval_fp8corr': (I, [ctypes.POINTER(Mlp), P, I64, P, P]),
    'nefii_sdf_fp8corr_supported': (I, [ctypes.POINTER(Mlp)]),
    'nefii_trace_workspace_bytes': (ctypes.c_size_t, [I64, ctypes.POINTER(TracerParams)]),
    'nefii_trace_max_rounds': (I, [ctypes.POINTER(TracerParams)]),
    'nefii_trace_rays': (I, [ctypes.POINTER(Mlp), ctypes.POINTER(TracerParams), P, P, P, I64, P, P, P, P, P, P,
                             ctypes.c_size_t, P, P]),
    'nefii_trace_rays_rounds': (I, [ctypes.POINTER(Mlp), ctypes.POINTER(TracerParams), P, P, P, I64, P, P, P, P, P, P,
                                    ctypes.c_size_t, P, I, I, P]),
    'nefii_trace_rays_groups': (I, [ctypes.POINTER(Mlp), ctypes.POINTER(TracerParams), P, P, P, I, P, P, P, P, P, P, P, P,
                                    P, I, I, P]),
    'nefii_trace_profile_enable': (I, [I]),
    'nefii_trace_profile_read': (I, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(I), ctypes.POINTER(ctypes.c_double)]),
    'nefii_trace_profile_launches': (I, [ctypes.POINTER(ctypes.c_float), I]),
    'nefii_idr_loss': (I, [ctypes.POINTER(LossParams), P, P, P, P, P, P, P, I64, P, P, P, P]),
    'nefii_camera_rays': (I, [P, P, P, I, I64, P, P, P]),
    'nefii_assemble_rows': (I, [ctypes.POINTER(RowBlock), I, P, I64, I64, P]),
    'nefii_gather_rows': (I, [ctypes.POINTER(RowBlock), I, P, I64, I64, P]),
    'nefii_prepare_hits': (I, [P, P, P, P, I, P, I64, I64, P, P, P, P, P]),
    'nefii_material_head_global': (I, [P, P, I, I, I, P, P, P]),
    'nefii_material_head_global_backward': (I, [P, P, I, I, I, P, P, P, P, P]),
    'nefii_sg_render_forward': (I, [P, I, P, P, P, P, P, I64, P, P, P, P]),
    'nefii_sg_render_backward': (I, [P, I, P, P, P, P, P, I64, P, P, P, P, P, P, P, P]),
    'nefii_env_radiance_forward': (I, [P, I, P, I64, F, P, P]),
    'nefii_env_radiance_backward': (I, [P, I, P, I64, F, P, P, P]),
    'nefii_mis_sample': (I, [P, I, P, P, P, P, I64, P, P, P, P]),
    'nefii_mc_shade_forward': (I, [P] * 11 + [I64, P, P, P, P]),
    'nefii_mc_shade_backward': (I, [P] * 11 + [I64] + [P] * 9),
    'nefii_mfma_sustained_probe': (I, [I, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), P]),
    'nefii_mfma_sustained_probe_chains': (I, [I, I, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), P]),
}

_lib = None


class NefiiLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NefiiLibraryError(
            'libnefii_hip.so is not built (%s missing). Run `python -m nefii_amd.build`. '
            'There is no CPU/PyTorch fallback for the hot path.' % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the one (and only) HIP runtime in the process,
    # so make sure torch has loaded it before our library resolves the same SONAME.
    import torch  # noqa: F401
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:          # e.g. libamdhip64 absent
        raise NefiiLibraryError('cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise NefiiLibraryError('%s does not export %s' % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if handle.nefii_abi_version() != ABI_VERSION:
        raise NefiiLibraryError('ABI version mismatch')
    _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed with status %d (%s)' % (
            what, rc, {-1: 'bad argument', -2: 'bad shape', -3: 'unsupported'}.get(rc, 'hipError_t')))
