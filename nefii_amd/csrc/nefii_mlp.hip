// nefii_mlp.hip - fused MLP kernels (forward, hidden-gradient backward, SDF value+gradient) and the
// weight packer, for gfx950.  Tile machinery in mlp_tile.h.
#include <cstdlib>

#include "mlp_tile.h"

using namespace nefii;

#define HIP_CHECK_LAUNCH()                       \
    do {                                         \
        hipError_t _e = hipGetLastError();       \
        if (_e != hipSuccess) return (int)_e;    \
    } while (0)

static inline int round32(int v) { return (v + 31) & ~31; }
// hidden / output widths: multiples of 64 once wider than one 32-column tile, so that every 16-deep k-step count of
// the split-precision kernels is a multiple of 4 (their 4-stage fragment pipeline never needs a remainder loop)
static inline int pad_hidden(int v) { return v <= 32 ? round32(v) : (v + 63) & ~63; }
extern "C" int nefii_padded_width(int v) { return v < 0 ? NEFII_E_ARG : pad_hidden(v); }

extern "C" int nefii_abi_version(void) { return NEFII_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int src_col(int kk, int kx, int x_src0, int x_len, int e_src0, int e_len) {
    if (kk < kx) return kk < x_len ? x_src0 + kk : -1;
    int e = kk - kx;
    return e < e_len ? e_src0 + e : -1;
}

__global__ void pack_linear_kernel(const float *__restrict__ W, const float *__restrict__ bias, int n_out, int k_in,
                                   int kx, int ke, int n_pad, int x_src0, int x_len, int e_src0, int e_len, float scale,
                                   float *__restrict__ w_fwd, float *__restrict__ w_bwd, float *__restrict__ bias_pad) {
    const int K = kx + ke;
    const int total = K * n_pad;
    const int NT = n_pad >> 5, KT = K >> 5;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int s = idx & 3, lane = (idx >> 2) & 63, blk = idx >> 8;
        {   // forward fragment: blk = g*NT + t ; element W[n = 32t + (lane&31)][k = 8g + 4(lane>>5) + s]
            const int t = blk % NT, g = blk / NT;
            const int n = 32 * t + (lane & 31), kk = 8 * g + 4 * (lane >> 5) + s;
            const int c = src_col(kk, kx, x_src0, x_len, e_src0, e_len);
            w_fwd[idx] = (n < n_out && c >= 0) ? W[(size_t)n * k_in + c] * scale : 0.f;
        }
        if (w_bwd) {   // backward fragment: contraction over n, output column kk: blk = g*KT + t
            const int t = blk % KT, g = blk / KT;
            const int kk = 32 * t + (lane & 31), n = 8 * g + 4 * (lane >> 5) + s;
            const int c = src_col(kk, kx, x_src0, x_len, e_src0, e_len);
            w_bwd[idx] = (n < n_out && c >= 0) ? W[(size_t)n * k_in + c] * scale : 0.f;
        }
        if (idx < n_pad) bias_pad[idx] = (bias && idx < n_out) ? bias[idx] : 0.f;
    }
}

extern "C" int nefii_pack_linear(const float *W, const float *bias, int n_out, int k_in, int x_src0, int x_len,
                                 int e_src0, int e_len, float scale, float *w_fwd, float *w_bwd, float *bias_pad,
                                 void *stream) {
    if (!W || !w_fwd || !bias_pad || n_out <= 0 || k_in <= 0) return NEFII_E_ARG;
    const int kx = pad_hidden(x_len), ke = round32(e_len), n_pad = pad_hidden(n_out);
    if (n_pad > NEFII_MAX_WIDTH || kx > NEFII_MAX_WIDTH || ke > NEFII_MAX_ENC || kx + ke == 0) return NEFII_E_SHAPE;
    if (x_src0 + x_len > k_in || e_src0 + e_len > k_in) return NEFII_E_SHAPE;
    const int total = (kx + ke) * n_pad;
    int blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_linear_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, bias, n_out, k_in, kx, ke,
                       n_pad, x_src0, x_len, e_src0, e_len, scale, w_fwd, w_bwd, bias_pad);
    HIP_CHECK_LAUNCH();
    return 0;
}

// fp16 hi/lo split in 32x32x16 fragment order: half8 index ((s*NT + t)*2 + part)*64 + lane, element j:
// W[n = 32t + (lane&31)][k = 16s + 8(lane>>5) + j] * scale * 64
__global__ void pack_linear_f16x3_kernel(const float *__restrict__ W, int n_out, int k_in, int kx, int ke, int n_pad,
                                         int x_src0, int x_len, int e_src0, int e_len, float scale,
                                         _Float16 *__restrict__ out) {
    const int K = kx + ke, NT = n_pad >> 5;
    const int total = (K >> 4) * NT * 64;          // (s, t, lane) triples
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, blk = idx >> 6;
        const int t = blk % NT, st = blk / NT;
        const int n = 32 * t + (lane & 31);
        _Float16 *hi = out + (((size_t)blk * 2) * 64 + lane) * 8;
        _Float16 *lo = out + (((size_t)blk * 2 + 1) * 64 + lane) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * st + 8 * (lane >> 5) + j;
            const int c = src_col(kk, kx, x_src0, x_len, e_src0, e_len);
            const float w = (n < n_out && c >= 0) ? W[(size_t)n * k_in + c] * scale * W16_SCALE : 0.f;
            split16(w, hi[j], lo[j]);
        }
    }
}

extern "C" int nefii_pack_linear_f16x3(const float *W, int n_out, int k_in, int x_src0, int x_len, int e_src0, int e_len,
                                       float scale, void *w_f16x3, void *stream) {
    if (!W || !w_f16x3 || n_out <= 0 || k_in <= 0) return NEFII_E_ARG;
    const int kx = pad_hidden(x_len), ke = round32(e_len), n_pad = pad_hidden(n_out);
    if (n_pad > NEFII_MAX_WIDTH || kx > NEFII_MAX_WIDTH || ke > NEFII_MAX_ENC || kx + ke == 0) return NEFII_E_SHAPE;
    if (x_src0 + x_len > k_in || e_src0 + e_len > k_in) return NEFII_E_SHAPE;
    const int total = ((kx + ke) >> 4) * (n_pad >> 5) * 64;
    int blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_linear_f16x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, n_out, k_in, kx, ke,
                       n_pad, x_src0, x_len, e_src0, e_len, scale, (_Float16 *)w_f16x3);
    HIP_CHECK_LAUNCH();
    return 0;
}

// transposed fragments for input-gradient GEMMs (dX = dZ * W): contraction over the outputs n, output column kk of the
// [X | E] input space.  half8 index ((s*KT + t)*2 + part)*64 + lane, element j:
// W[n = 16s + 8(lane>>5) + j][src_col(kk = 32t + (lane&31))] * scale * 64
__global__ void pack_linear_f16x3_bwd_kernel(const float *__restrict__ W, int n_out, int k_in, int kx, int ke, int n_pad,
                                             int x_src0, int x_len, int e_src0, int e_len, float scale,
                                             _Float16 *__restrict__ out) {
    const int KT = (kx + ke) >> 5;
    const int total = (n_pad >> 4) * KT * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, blk = idx >> 6;
        const int t = blk % KT, st = blk / KT;
        const int kk = 32 * t + (lane & 31);
        const int c = src_col(kk, kx, x_src0, x_len, e_src0, e_len);
        _Float16 *hi = out + (((size_t)blk * 2) * 64 + lane) * 8;
        _Float16 *lo = out + (((size_t)blk * 2 + 1) * 64 + lane) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int nn = 16 * st + 8 * (lane >> 5) + j;
            const float w = (nn < n_out && c >= 0) ? W[(size_t)nn * k_in + c] * scale * W16_SCALE : 0.f;
            split16(w, hi[j], lo[j]);
        }
    }
}

extern "C" int nefii_pack_linear_f16x3_bwd(const float *W, int n_out, int k_in, int x_src0, int x_len, int e_src0,
                                           int e_len, float scale, void *w_bwd_f16x3, void *stream) {
    if (!W || !w_bwd_f16x3 || n_out <= 0 || k_in <= 0) return NEFII_E_ARG;
    const int kx = pad_hidden(x_len), ke = round32(e_len), n_pad = pad_hidden(n_out);
    if (n_pad > NEFII_MAX_WIDTH || kx > NEFII_MAX_WIDTH || ke > NEFII_MAX_ENC || kx + ke == 0) return NEFII_E_SHAPE;
    if (x_src0 + x_len > k_in || e_src0 + e_len > k_in) return NEFII_E_SHAPE;
    // n_pad is 32 for the one-output last layer: its 16-deep k-steps still come in pairs
    const int total = (n_pad >> 4) * ((kx + ke) >> 5) * 64;
    int blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_linear_f16x3_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, n_out, k_in, kx,
                       ke, n_pad, x_src0, x_len, e_src0, e_len, scale, (_Float16 *)w_bwd_f16x3);
    HIP_CHECK_LAUNCH();
    return 0;
}

// ---- the three packings above for EVERY layer of a net in one launch (blockIdx.y = layer) -------------------------------
// The radiance / material weights train, so they are re-packed every step: layer by layer that was 3 launches x 5 layers
// x 2 nets of ~4 us each at the head of the step's tail.
struct PackSources {
    nefii_pack_source l[NEFII_MAX_LAYERS];
};
__global__ void pack_mlp_kernel(nefii_mlp m, PackSources src) {
    const int l = blockIdx.y;
    const nefii_layer &L = m.layer[l];
    const nefii_pack_source &S = src.l[l];
    const float *__restrict__ W = S.W;
    const int kx = L.k_x, ke = L.k_e, n_pad = L.n_pad, K = kx + ke, n_out = S.n_out, k_in = S.k_in;
    const int NT = n_pad >> 5, KT = K >> 5;
    const float scale = S.scale;
    const int stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
    float *w_fwd = S.skip_f32 ? nullptr : const_cast<float *>(L.w_fwd);
    float *w_bwd = S.skip_f32 ? nullptr : const_cast<float *>(L.w_bwd);
    float *bias_pad = const_cast<float *>(L.bias);
    for (int idx = first; idx < n_pad; idx += stride) bias_pad[idx] = (S.bias && idx < n_out) ? S.bias[idx] : 0.f;
    if (w_fwd) {
        const int total = K * n_pad;
        for (int idx = first; idx < total; idx += stride) {
            const int s = idx & 3, lane = (idx >> 2) & 63, blk = idx >> 8;
            {
                const int t = blk % NT, g = blk / NT;
                const int n = 32 * t + (lane & 31), kk = 8 * g + 4 * (lane >> 5) + s;
                const int c = src_col(kk, kx, S.x_src0, S.x_len, S.e_src0, S.e_len);
                w_fwd[idx] = (n < n_out && c >= 0) ? W[(size_t)n * k_in + c] * scale : 0.f;
            }
            if (w_bwd) {
                const int t = blk % KT, g = blk / KT;
                const int kk = 32 * t + (lane & 31), n = 8 * g + 4 * (lane >> 5) + s;
                const int c = src_col(kk, kx, S.x_src0, S.x_len, S.e_src0, S.e_len);
                w_bwd[idx] = (n < n_out && c >= 0) ? W[(size_t)n * k_in + c] * scale : 0.f;
            }
        }
    }
    if (L.w_f16x3) {
        _Float16 *out = reinterpret_cast<_Float16 *>(const_cast<void *>(L.w_f16x3));
        const int total = (K >> 4) * NT * 64;
        for (int idx = first; idx < total; idx += stride) {
            const int lane = idx & 63, blk = idx >> 6;
            const int t = blk % NT, st = blk / NT;
            const int n = 32 * t + (lane & 31);
            _Float16 *hi = out + (((size_t)blk * 2) * 64 + lane) * 8, *lo = out + (((size_t)blk * 2 + 1) * 64 + lane) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = 16 * st + 8 * (lane >> 5) + j;
                const int c = src_col(kk, kx, S.x_src0, S.x_len, S.e_src0, S.e_len);
                const float w = (n < n_out && c >= 0) ? W[(size_t)n * k_in + c] * scale * W16_SCALE : 0.f;
                split16(w, hi[j], lo[j]);
            }
        }
    }
    if (L.w_bwd_f16x3) {
        _Float16 *out = reinterpret_cast<_Float16 *>(const_cast<void *>(L.w_bwd_f16x3));
        const int total = (n_pad >> 4) * KT * 64;
        for (int idx = first; idx < total; idx += stride) {
            const int lane = idx & 63, blk = idx >> 6;
            const int t = blk % KT, st = blk / KT;
            const int kk = 32 * t + (lane & 31);
            const int c = src_col(kk, kx, S.x_src0, S.x_len, S.e_src0, S.e_len);
            _Float16 *hi = out + (((size_t)blk * 2) * 64 + lane) * 8, *lo = out + (((size_t)blk * 2 + 1) * 64 + lane) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int nn = 16 * st + 8 * (lane >> 5) + j;
                const float w = (nn < n_out && c >= 0) ? W[(size_t)nn * k_in + c] * scale * W16_SCALE : 0.f;
                split16(w, hi[j], lo[j]);
            }
        }
    }
}

extern "C" int nefii_pack_mlp(const nefii_mlp *h_mlp, const nefii_pack_source *h_layers, void *stream) {
    if (!h_mlp || !h_layers || h_mlp->n_layers < 1 || h_mlp->n_layers > NEFII_MAX_LAYERS) return NEFII_E_ARG;
    PackSources src;
    for (int l = 0; l < h_mlp->n_layers; ++l) {
        const nefii_layer &L = h_mlp->layer[l];
        const nefii_pack_source &S = h_layers[l];
        if (!S.W || !L.bias || (!S.skip_f32 && !L.w_fwd) || S.n_out <= 0 || S.k_in <= 0) return NEFII_E_ARG;
        if (L.k_x != pad_hidden(S.x_len) || L.k_e != round32(S.e_len) || L.n_pad != pad_hidden(S.n_out)) return NEFII_E_SHAPE;
        if (L.n_pad > NEFII_MAX_WIDTH || L.k_x > NEFII_MAX_WIDTH || L.k_e > NEFII_MAX_ENC || L.k_x + L.k_e == 0) return NEFII_E_SHAPE;
        if (S.x_src0 + S.x_len > S.k_in || S.e_src0 + S.e_len > S.k_in) return NEFII_E_SHAPE;
        src.l[l] = S;
    }
    hipLaunchKernelGGL(pack_mlp_kernel, dim3(64, h_mlp->n_layers), dim3(256), 0, (hipStream_t)stream, *h_mlp, src);
    HIP_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// shared prologue: stage raw inputs + features of one 32-point tile, encode into E
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_tile_inputs(const nefii_mlp &m, const float *in_a, const float *in_b,
                                                 const float *in_c, const float *feat, int64_t base, int64_t n,
                                                 float *raw, Lds &lds) {
    const int tid = threadIdx.x;
    for (int i = tid; i < TILE * 9; i += WG) {     // 288 entries > 256 threads
        const int p = i / 9, c = i - 9 * p, which = c / 3;
        const float *src = which == 0 ? in_a : (which == 1 ? in_b : in_c);
        int64_t idx = base + p;
        if (idx >= n) idx = n - 1;
        raw[i] = (src && m.enc_freqs[which] >= 0) ? src[idx * 3 + (c - 3 * which)] : 0.f;
    }
    const int F = m.feat_width;
    const int kx0 = m.layer[0].k_x;
    if (kx0 > 0) {
        for (int i = tid; i < TILE * kx0; i += WG) {
            const int p = i / kx0, f = i - p * kx0;
            int64_t idx = base + p;
            if (idx >= n) idx = n - 1;
            lds.X[p * XS + f] = (feat && f < F) ? feat[idx * F + f] : 0.f;
        }
    }
    __syncthreads();
    int ke = 0;
    for (int l = 0; l < m.n_layers; ++l) ke = m.layer[l].k_e > ke ? m.layer[l].k_e : ke;
    encode_tile(m, raw, lds.E, ke);
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// one workgroup per CU: the epilogue (stash + head functions) needs more than 256 registers next to the pipeline
__global__ __launch_bounds__(256, 1) void mlp_forward_kernel(nefii_mlp m, const float *__restrict__ in_a,
                                                             const float *__restrict__ in_b,
                                                             const float *__restrict__ in_c,
                                                             const float *__restrict__ feat, int64_t n,
                                                             float *__restrict__ out, int out_stride,
                                                             float *__restrict__ hidden_out, int hid_stride,
                                                             float *__restrict__ stash, int stash_stride) {
    __shared__ Lds lds;
    __shared__ float raw[TILE * 9];
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * TILE;
        load_tile_inputs(m, in_a, in_b, in_c, feat, base, n, raw, lds);
        for (int l = 0; l < m.n_layers; ++l) {
            const nefii_layer &L = m.layer[l];
            f32x16 acc[4];
            int ntw;
            layer_gemm(L, lds.X, lds.E, L.w_fwd, L.n_pad >> 5, acc, ntw);
            __syncthreads();
            const bool last = (l == m.n_layers - 1);
            const bool pre_last = (l == m.n_layers - 2);
            NEFII_ACT_SWITCH(m.act, {
                NEFII_FOR_ACC(acc, ntw, {
                    const float z = val + L.bias[col];
                    const bool live = (base + row) < n;
                    if (!last) {
                        const float hval = act_fwd(z, ACT);
                        lds.X[row * XS + col] = hval;
                        if (stash && live) stash[((size_t)l * n + base + row) * stash_stride + col] = hval;
                        if (pre_last && hidden_out && live && col < L.n_out)
                            hidden_out[(size_t)(base + row) * hid_stride + col] = hval;
                    } else {
                        if (stash && live) stash[((size_t)l * n + base + row) * stash_stride + col] = z;
                        if (live && col < L.n_out) out[(size_t)(base + row) * out_stride + col] = head_fwd(z, m.head);
                    }
                })
            })
            __syncthreads();
        }
    }
}

// need_f32: the caller reads the f32 fragments (w_fwd); the fp16-MFMA entry points do not - nets that only run on those
// carry NULL there (nefii_amd.ops.PackedMLP(half=...)), so that an f32 entry point called on them fails instead of
// multiplying by weights that were never packed
static int check_mlp(const nefii_mlp *m, bool need_f32 = true) {
    if (!m || m->n_layers < 1 || m->n_layers > NEFII_MAX_LAYERS) return NEFII_E_ARG;
    for (int l = 0; l < m->n_layers; ++l) {
        const nefii_layer &L = m->layer[l];
        if ((L.k_x & 31) || (L.k_e & 31) || (L.n_pad & 31) || L.k_x + L.k_e <= 0) return NEFII_E_SHAPE;
        if (L.k_x > NEFII_MAX_WIDTH || L.k_e > NEFII_MAX_ENC || L.n_pad > NEFII_MAX_WIDTH || L.n_out > L.n_pad)
            return NEFII_E_SHAPE;
        if ((need_f32 && !L.w_fwd) || !L.bias) return NEFII_E_ARG;
        if (l > 0 && L.k_x != m->layer[l - 1].n_pad) return NEFII_E_SHAPE;
    }
    int ew = 0;
    for (int i = 0; i < 3; ++i)
        if (m->enc_freqs[i] >= 0) ew += 3 + 6 * m->enc_freqs[i];
    if (ew > NEFII_MAX_ENC) return NEFII_E_SHAPE;
    if (m->feat_width > m->layer[0].k_x) return NEFII_E_SHAPE;
    return 0;
}

static int grid_for(int64_t n_tiles, int per_cu) {
    int64_t cap = 256 * per_cu * 2;   // 256 CUs; tiles beyond the cap are grid-strided
    return (int)(n_tiles < cap ? (n_tiles > 0 ? n_tiles : 1) : cap);
}

extern "C" int nefii_mlp_forward(const nefii_mlp *h_mlp, const float *in_a, const float *in_b, const float *in_c,
                                 const float *feat, int64_t n, float *out, int out_stride, float *hidden_out,
                                 int hid_stride, float *stash, int stash_stride, void *stream) {
    int rc = check_mlp(h_mlp);
    if (rc) return rc;
    if (n <= 0) return 0;
    if (!out) return NEFII_E_ARG;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    hipLaunchKernelGGL(mlp_forward_kernel, dim3(grid_for(n_tiles, 1)), dim3(WG), 0, (hipStream_t)stream, *h_mlp, in_a,
                       in_b, in_c, feat, n, out, out_stride, hidden_out, hid_stride, stash, stash_stride);
    HIP_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// encode only: the reference's layer-0 concatenation [PE(a) | PE(b) | PE(c) | feat] as a dense matrix
// (operand of the layer-0 weight-gradient GEMM)
// ------------------------------------------------------------------------------------------------
__global__ void encode_kernel(nefii_mlp m, const float *__restrict__ in_a, const float *__restrict__ in_b,
                              const float *__restrict__ in_c, const float *__restrict__ feat, int64_t n,
                              float *__restrict__ out, int width) {
    const int w0 = enc_width(m.enc_freqs[0]), w1 = enc_width(m.enc_freqs[1]), w2 = enc_width(m.enc_freqs[2]);
    const int we = w0 + w1 + w2, F = m.feat_width;
    const int64_t total = n * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / width;
        const int c = (int)(i - p * width);
        float v[3];
        float val;
        if (c < we) {
            const float *src = c < w0 ? in_a : (c < w0 + w1 ? in_b : in_c);
            const int cc = c < w0 ? c : (c < w0 + w1 ? c - w0 : c - w0 - w1);
            v[0] = src[p * 3], v[1] = src[p * 3 + 1], v[2] = src[p * 3 + 2];
            val = enc_value(v, cc);
        } else {
            val = (c - we) < F ? feat[p * F + (c - we)] : 0.f;
        }
        out[i] = val;
    }
}

extern "C" int nefii_encode_inputs(const nefii_mlp *h_mlp, const float *in_a, const float *in_b, const float *in_c,
                                   const float *feat, int64_t n, float *out, int width, void *stream) {
    if (!h_mlp || !out) return NEFII_E_ARG;
    if (n <= 0) return 0;
    int64_t total = n * width;
    int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(encode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *h_mlp, in_a, in_b, in_c, feat,
                       n, out, width);
    HIP_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward wrt hidden activations (parameters get their gradients from dz via a GEMM per layer)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void mlp_backward_kernel(nefii_mlp m, const float *__restrict__ d_out,
                                                              int out_stride, const float *__restrict__ stash,
                                                              int stash_stride, int64_t n, float *__restrict__ dz,
                                                              int dz_stride) {
    __shared__ Lds lds;
    const int tid = threadIdx.x;
    const int Lm1 = m.n_layers - 1;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * TILE;
        {   // seed: dz_{L-1} = d_out * head'(pre)
            const nefii_layer &L = m.layer[Lm1];
            for (int i = tid; i < TILE * L.n_pad; i += WG) {
                const int p = i / L.n_pad, c = i - p * L.n_pad;
                float v = 0.f;
                if (base + p < n && c < L.n_out) {
                    const float pre = stash[((size_t)Lm1 * n + base + p) * stash_stride + c];
                    const float y = head_fwd(pre, m.head);
                    v = d_out[(size_t)(base + p) * out_stride + c] * head_bwd_from_out(y, pre, m.head);
                }
                lds.X[p * XS + c] = v;
                if (base + p < n) dz[((size_t)Lm1 * n + base + p) * dz_stride + c] = v;
            }
        }
        __syncthreads();
        for (int l = Lm1; l >= 1; --l) {
            const nefii_layer &L = m.layer[l];
            // dH_{l-1}[32 x k_x] = dZ_l[32 x n_pad] * W_l   (only the hidden block of the inputs)
            f32x16 acc[4];
            int ntw;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            const int NTs = (L.k_x + L.k_e) >> 5, ntc = L.k_x >> 5;
            ntw = (ntc - wave + 3) >> 2;
            if (ntw < 0) ntw = 0;
            zero_acc(acc);
            gemm_block(lds.X, XS, L.n_pad >> 3, reinterpret_cast<const float4 *>(L.w_bwd), NTs, wave, lane, ntw, acc);
            __syncthreads();
            NEFII_ACT_SWITCH(m.act, {
                NEFII_FOR_ACC(acc, ntw, {
                    const bool live = (base + row) < n;
                    float v = 0.f;
                    if (live) {
                        const float hprev = stash[((size_t)(l - 1) * n + base + row) * stash_stride + col];
                        v = val * act_bwd_from_out(hprev, ACT);
                        dz[((size_t)(l - 1) * n + base + row) * dz_stride + col] = v;
                    }
                    lds.X[row * XS + col] = v;
                })
            })
            __syncthreads();
        }
    }
}

extern "C" int nefii_mlp_backward(const nefii_mlp *h_mlp, const float *d_out, int out_stride, const float *stash,
                                  int stash_stride, int64_t n, float *dz, int dz_stride, void *stream) {
    int rc = check_mlp(h_mlp);
    if (rc) return rc;
    if (n <= 0) return 0;
    if (!d_out || !stash || !dz) return NEFII_E_ARG;
    for (int l = 1; l < h_mlp->n_layers; ++l)
        if (!h_mlp->layer[l].w_bwd) return NEFII_E_ARG;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    hipLaunchKernelGGL(mlp_backward_kernel, dim3(grid_for(n_tiles, 2)), dim3(WG), 0, (hipStream_t)stream, *h_mlp, d_out,
                       out_stride, stash, stash_stride, n, dz, dz_stride);
    HIP_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight / bias gradients:  dW[n][k] = scale * sum_p dz[p][n] * x[p][k] ,  db[n] = sum_p dz[p][n]
// One wave = a 32 x 128 strip of dW (4 accumulator tiles); the contraction runs over the points, two per
// v_mfma_f32_32x32x2_f32 (A = dz^T fragment: lane (n, p&1), B = x fragment: lane (k, p&1), both 128-B coalesced rows).
// Workgroup = 64 x 256 block; the point range is split over gridDim.z workgroups that combine with fp32 atomics
// (few: <= 32 adders per element).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_wgrad_kernel(const float *__restrict__ dz, int dz_stride,
                                                        const float *__restrict__ x, int x_stride, int64_t P, int n_out,
                                                        int k_in, float scale, float *__restrict__ dW,
                                                        float *__restrict__ db, int atomic) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 64 + (wave & 1) * 32;
    const int k0 = blockIdx.y * 256 + (wave >> 1) * 128;
    const int64_t chunk = ((P + gridDim.z - 1) / gridDim.z + 1) & ~(int64_t)1;
    const int64_t p_begin = (int64_t)blockIdx.z * chunk;
    const int64_t p_end = p_begin + chunk < P ? p_begin + chunk : P;
    f32x16 acc[4];
    zero_acc(acc);
    const bool n_ok = (n0 + i) < n_out;
    bool k_ok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) k_ok[t] = (k0 + 32 * t + i) < k_in;
    float bsum = 0.f;
    if (n0 < n_out && k0 < k_in) {
        constexpr int U = 8;                       // point pairs per batch: all loads of a batch are issued before its MFMAs
        for (int64_t pb = p_begin; pb < p_end; pb += 2 * U) {
            float a[U], b[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t p = pb + 2 * u + h;
                const bool p_ok = p < p_end;
                a[u] = (p_ok && n_ok) ? dz[p * dz_stride + n0 + i] : 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) b[u][t] = (p_ok && k_ok[t]) ? x[p * x_stride + k0 + 32 * t + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                bsum += a[u];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * h, k = k0 + 32 * t + i;
                if (n < n_out && k < k_in) {
                    const float v = acc[t][r] * scale;
                    if (atomic) atomicAdd(&dW[(size_t)n * k_in + k], v);
                    else dW[(size_t)n * k_in + k] = v;
                }
            }
    }
    if (db && blockIdx.y == 0 && (wave >> 1) == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0 && n_ok) {
            if (atomic) atomicAdd(&db[n0 + i], bsum);
            else db[n0 + i] = bsum;
        }
    }
}

// zeroes dW [nw] and, behind it in the same grid, db [nb] (may be NULL): one launch ahead of an accumulating wgrad kernel
__global__ void zero_fill_kernel(float *__restrict__ p, size_t n, float *__restrict__ q, size_t nq) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
    else if (q && i - n < nq) q[i - n] = 0.f;
}

extern "C" int nefii_mlp_wgrad(const float *dz, int dz_stride, const float *x, int x_stride, int64_t n, int n_out,
                               int k_in, float scale, float *dW, float *db, void *stream) {
    if (!dz || !x || !dW || n_out <= 0 || k_in <= 0) return NEFII_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    int split = (int)((n + 127) / 128);       // <= 128 points per workgroup: the kernel is load-latency bound
    if (split < 1) split = 1;
    if (split > 32) split = 32;
    if (split > 1 || n <= 0) {
        // a kernel, not hipMemsetAsync: inside a captured hipGraph (TrainStep(graph=True)) the memset node did not
        // reliably precede the accumulating kernel on replay - weight gradients picked up non-finite garbage
        const size_t nw = (size_t)n_out * k_in;
        const size_t nz = nw + (db ? (size_t)n_out : 0);
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, dW, nw, db, (size_t)n_out);
        HIP_CHECK_LAUNCH();
    }
    if (n <= 0) return 0;
    dim3 grid((n_out + 63) / 64, (k_in + 255) / 256, split);
    hipLaunchKernelGGL(mlp_wgrad_kernel, grid, dim3(256), 0, st, dz, dz_stride, x, x_stride, n, n_out, k_in, scale, dW, db,
                       split > 1 ? 1 : 0);
    HIP_CHECK_LAUNCH();
    return 0;
}

// ================================================================================================
// Single-pass fp16 variants for the radiance and material MLPs (RenderingNetwork :196-241, EnvmapMaterialNetwork's
// diffuse_albedo_layers sg_envmap_material.py:369): operands rounded to fp16 once, one v_mfma_f32_32x32x16_f16 per
// 16-deep k-step, fp32 accumulation, fp32 bias / activation / head - 16x the f32-input MFMA rate of the kernels above,
// at which these became 14 % of a config-3 step.  SURVEY.md section 7 measured fp16 on exactly these two nets at
// 1.9e-4 relative L2 on rendered RGB (north-star tolerance 1e-3); the SDF network never takes this path.  Weights are
// the hi blocks of nefii_pack_linear_f16x3 / _bwd (x 64), activations carry A16_SCALE.  Gradients are far below fp16's
// normal range (d loss / d rgb ~ 1e-6 per ray), so the backward GEMMs carry dz x S, S a power of two chosen per call
// from max |d_out| (nefii_mlp_grad_scale); everything leaves the kernels in fp32, unscaled.
// ================================================================================================
// LDS image of a 32-row tile: hi halves (single pass) or hi + lo halves (split precision) of activations / encoded inputs
template <bool SP>
struct LdsF16 {
    _Float16 Xh[TILE * XS16], Eh[TILE * ES16];
    _Float16 Xl[SP ? 8 : TILE * XS16], El[SP ? 8 : TILE * ES16];
    __device__ __forceinline__ void put_x(int idx, float v) {
        const _Float16 hi = (_Float16)(v * A16_SCALE);
        Xh[idx] = hi;
        if constexpr (!SP) Xl[idx] = (_Float16)(v * A16_SCALE - (float)hi);
    }
    __device__ __forceinline__ void put_e(int idx, float v) {
        const _Float16 hi = (_Float16)(v * A16_SCALE);
        Eh[idx] = hi;
        if constexpr (!SP) El[idx] = (_Float16)(v * A16_SCALE - (float)hi);
    }
};

template <bool SP>
__device__ __forceinline__ void load_tile_inputs16h(const nefii_mlp &m, const float *in_a, const float *in_b,
                                                    const float *in_c, const float *feat, int64_t base, int64_t n,
                                                    float *raw, LdsF16<SP> &lds) {
    const int tid = threadIdx.x;
    for (int i = tid; i < TILE * 9; i += WG) {
        const int p = i / 9, c = i - 9 * p, which = c / 3;
        const float *src = which == 0 ? in_a : (which == 1 ? in_b : in_c);
        int64_t idx = base + p;
        if (idx >= n) idx = n - 1;
        raw[i] = (src && m.enc_freqs[which] >= 0) ? src[idx * 3 + (c - 3 * which)] : 0.f;
    }
    const int F = m.feat_width;
    const int kx0 = m.layer[0].k_x;
    if (kx0 > 0) {
        for (int i = tid; i < TILE * kx0; i += WG) {
            const int p = i / kx0, f = i - p * kx0;
            int64_t idx = base + p;
            if (idx >= n) idx = n - 1;
            lds.put_x(p * XS16 + f, (feat && f < F) ? feat[idx * F + f] : 0.f);
        }
    }
    __syncthreads();
    int ke = 0;
    for (int l = 0; l < m.n_layers; ++l) ke = m.layer[l].k_e > ke ? m.layer[l].k_e : ke;
    const int pp = tid & 31, part = tid >> 5;
    const int w0 = enc_width(m.enc_freqs[0]), w1 = enc_width(m.enc_freqs[1]), w2 = enc_width(m.enc_freqs[2]);
    for (int c = part; c < ke; c += 8) {
        float val = 0.f;
        if (c < w0) val = enc_value(raw + pp * 9, c);
        else if (c < w0 + w1) val = enc_value(raw + pp * 9 + 3, c - w0);
        else if (c < w0 + w1 + w2) val = enc_value(raw + pp * 9 + 6, c - w0 - w1);
        lds.put_e(pp * ES16 + c, val);
    }
    __syncthreads();
}

// SP = false: split precision (fp16 hi/lo pairs, 3 MFMAs per k-step: fp32-class accuracy - the default for the forward
// pass, whose outputs are held to the north-star tolerance); SP = true: one fp16 pass.
template <bool SP>
__global__ __launch_bounds__(256, 1) void mlp_forward16_kernel(nefii_mlp m, const float *__restrict__ in_a,
                                                               const float *__restrict__ in_b,
                                                               const float *__restrict__ in_c,
                                                               const float *__restrict__ feat, int64_t n,
                                                               float *__restrict__ out, int out_stride,
                                                               float *__restrict__ hidden_out, int hid_stride,
                                                               float *__restrict__ stash, int stash_stride) {
    __shared__ LdsF16<SP> lds;
    __shared__ float raw[TILE * 9];
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * TILE;
        load_tile_inputs16h<SP>(m, in_a, in_b, in_c, feat, base, n, raw, lds);
        for (int l = 0; l < m.n_layers; ++l) {
            const nefii_layer &L = m.layer[l];
            f32x16 acc[4];
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
            const int n_tiles_l = L.n_pad >> 5;
            int ntw = (n_tiles_l - wave + 3) >> 2;
            if (ntw < 0) ntw = 0;
            zero_acc(acc);
            const half8 *wp = reinterpret_cast<const half8 *>(L.w_f16x3);
            gemm_block16<SP>(lds.Xh, lds.Xl, XS16, L.k_x >> 4, wp, n_tiles_l, wave, lane, ntw, acc);
            gemm_block16<SP>(lds.Eh, lds.El, ES16, L.k_e >> 4, wp + (size_t)(L.k_x >> 4) * n_tiles_l * 2 * 64, n_tiles_l,
                             wave, lane, ntw, acc);
            __syncthreads();
            const bool last = (l == m.n_layers - 1);
            const bool pre_last = (l == m.n_layers - 2);
            NEFII_ACT_SWITCH(m.act, {
                NEFII_FOR_ACC(acc, ntw, {
                    const float z = val * inv_scale + L.bias[col];
                    const bool live = (base + row) < n;
                    if (!last) {
                        const float hval = act_fwd(z, ACT);
                        lds.put_x(row * XS16 + col, hval);
                        if (stash && live) stash[((size_t)l * n + base + row) * stash_stride + col] = hval;
                        if (pre_last && hidden_out && live && col < L.n_out)
                            hidden_out[(size_t)(base + row) * hid_stride + col] = hval;
                    } else {
                        if (stash && live) stash[((size_t)l * n + base + row) * stash_stride + col] = z;
                        if (live && col < L.n_out) out[(size_t)(base + row) * out_stride + col] = head_fwd(z, m.head);
                    }
                })
            })
            __syncthreads();
        }
    }
}

// ================================================================================================
// Streamed split-precision forward for the radiance / material MLPs: the tracer's "16q" machine (mlp_tile.h) on nets
// whose hidden layers are all 512 wide - 8 waves, each owning 64 features of every hidden layer, the layers' hi/lo
// fragments read as ONE prefetched stream per wave (nefii_mlp.w_stream, nefii_pack_mlp_stream), 48- or 64-row tiles.
// The 32-row kernel above re-reads every layer's fragments per tile with nothing in flight: 9.3 us per row on
// config 3's secondary points against ~2 us per row here.
//   image row: [features of layer 0 | hidden block .. column 512) [encodings, zero-padded to EW) ; a layer's inputs are
//   the columns [512 - k_x, 512 + k_e) and its stream units cover that K rounded up to a multiple of 64 (zero weights).
//   64-row tiles; EW = 64: encodings of up to 64 columns (material: PE(x)); EW = 96: up to 96 (radiance: PE(x), PE(v),
//   n) - 158 of the 160 KiB of LDS.  The K padding may reach past a row's EW columns (radiance layer 0: 608 -> 640):
//   what it reads there - the next row's first columns, the tail behind the last row - is finite and meets zero weights.
// The last layer (n_pad 32, at most 8 outputs) is a 32x32x16 GEMM with K split over the waves, reduced through LDS.
// ================================================================================================
// ELU on the hardware exponential: e^x - 1 from v_exp_f32 (absolute error ~6e-8, the rounding of e^x near 1), and the
// degree-4 series where that cancels (|x| < 1/16: truncation error < 1e-8).  The libm expm1f of act_fwd is ~40
// instructions per value - on 64 values per lane and layer it was a third of the streamed tile's time.
__device__ __forceinline__ float elu_fast(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 1.44269504088896340736f) - 1.f;
    const float p = x * __builtin_fmaf(x, __builtin_fmaf(x, __builtin_fmaf(x, 1.f / 24.f, 1.f / 6.f), 0.5f), 1.f);
    const float r = __builtin_fabsf(x) < 0.0625f ? p : t;
    return x > 0.f ? x : r;
}

template <int QT, int EW>
struct LdsM {
    static constexpr int XP = 512 + EW + 8;         // halves per row: 584 / 616 (16 B * odd)
    _Float16 Xh[16 * QT * XP], Xl[16 * QT * XP];
    _Float16 tail[64];                              // the K padding and the A-fragment read-ahead of the last row land here
};
__host__ __device__ __forceinline__ int m_units(const nefii_layer &L) { return ((L.k_x + L.k_e + 63) & ~63) >> 4; }

// encoding width the streamed kernel needs for this net (64 / 96), 0: the net does not take it
static int mstream_shape(const nefii_mlp *m) {
    const int NH = m->n_layers - 1;
    if (NH < 1 || NH > 12) return 0;
    for (int l = 0; l <= NH; ++l)
        if (!m->layer[l].w_f16x3 || !m->layer[l].bias) return 0;
    const nefii_layer &L0 = m->layer[0];
    if (L0.n_pad != 512 || L0.k_x > 512 || (L0.k_x & 15) || L0.k_e > 96 || L0.k_x + L0.k_e == 0) return 0;
    for (int l = 1; l < NH; ++l)
        if (m->layer[l].n_pad != 512 || m->layer[l].k_x != 512 || m->layer[l].k_e != 0) return 0;
    const nefii_layer &LL = m->layer[NH];
    if (LL.k_x != 512 || LL.k_e != 0 || LL.n_pad != 32 || LL.n_out > 8) return 0;
    return L0.k_e <= 64 ? 64 : 96;
}
static int mstream_units(const nefii_mlp *m) {
    int G = 0;
    for (int l = 0; l < m->n_layers - 1; ++l) G += m_units(m->layer[l]);
    return G;
}

// backward copy (layers with transposed fragments): for l = NL-1 .. 1 the 32-deep k-steps of the transposed layer -
// contraction over layer l's outputs, padded to a multiple of 128 - hi fragments of the wave's 4 feature tiles (its 64 of
// the layer's 512 hidden inputs); the one-pass fp16 backward (mlp_backward16s_kernel) reads it
__host__ __device__ __forceinline__ int mb_units(const nefii_layer &L) { return ((L.n_pad + 127) & ~127) >> 5; }
static int mstream_units_bwd(const nefii_mlp *m) {
    int G = 0;
    for (int l = 1; l < m->n_layers; ++l) {
        if (!m->layer[l].w_bwd_f16x3) return 0;
        G += mb_units(m->layer[l]);
    }
    return G;
}

extern "C" size_t nefii_mlp_stream_bytes(const nefii_mlp *h_mlp) {
    if (!h_mlp || h_mlp->n_layers < 2 || h_mlp->n_layers > NEFII_MAX_LAYERS || !mstream_shape(h_mlp)) return 0;
    return (size_t)8 * (mstream_units(h_mlp) + mstream_units_bwd(h_mlp)) * 256 * sizeof(half8);
}

// dst[((wave*G + g)*4 + 2 f + part)*64 + lane][j] = W[n = 64 wave + 16 (2 (g&1) + f) + (lane&15)][k = 32 (g>>1) + 8 (lane>>4) + j]
// of unit g's layer (hi / lo), gathered from the layer's 32x32x16 fragments; zeros past the layer's own K
__global__ void pack_mlp_stream_kernel(nefii_mlp m, half8 *__restrict__ dst, int G) {
    const int g = blockIdx.x, wave = blockIdx.y;
    int l = 0, s = g;
    while (s >= m_units(m.layer[l])) s -= m_units(m.layer[l]), ++l;
    const half8 *w = reinterpret_cast<const half8 *>(m.layer[l].w_f16x3);
    const int frag = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = frag >> 1, part = frag & 1, kg = lane >> 4;
    const int half = s & 1, s32 = s >> 1;
    const int n = 64 * wave + 16 * (2 * half + f) + (lane & 15);
    const int s16 = 2 * s32 + (kg >> 1), t = n >> 5, lane_src = (n & 31) + 32 * (kg & 1);
    half8 v;
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
    if (s16 < ((m.layer[l].k_x + m.layer[l].k_e) >> 4)) v = w[(((size_t)s16 * 16 + t) * 2 + part) * 64 + lane_src];
    dst[((size_t)wave * G + g) * 256 + threadIdx.x] = v;
}

// dst[((wave*G + g)*4 + f)*64 + lane][j] = hi(W_l[n = 32 s + 8 (lane>>4) + j][k = 64 wave + 16 f + (lane&15)]) for unit g =
// (layer l counted down from the last, 32-deep step s); zeros past the layer's own outputs
__global__ void pack_mlp_stream_bwd_kernel(nefii_mlp m, half8 *__restrict__ dst, int G) {
    const int g = blockIdx.x, wave = blockIdx.y;
    int l = m.n_layers - 1, s = g;
    while (s >= mb_units(m.layer[l])) s -= mb_units(m.layer[l]), --l;
    const nefii_layer &L = m.layer[l];
    const half8 *wb = reinterpret_cast<const half8 *>(L.w_bwd_f16x3);
    const int KT = (L.k_x + L.k_e) >> 5;
    const int f = threadIdx.x >> 6, lane = threadIdx.x & 63, kg = lane >> 4;
    const int kk = 64 * wave + 16 * f + (lane & 15);
    const int s16 = 2 * s + (kg >> 1), t = kk >> 5, lane_src = (kk & 31) + 32 * (kg & 1);
    half8 v;
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
    if (s16 < (L.n_pad >> 4)) v = wb[(((size_t)s16 * KT + t) * 2) * 64 + lane_src];
    dst[((size_t)wave * G + g) * 256 + threadIdx.x] = v;
}

extern "C" int nefii_pack_mlp_stream(const nefii_mlp *h_mlp, void *w_stream, void *stream) {
    if (!h_mlp || !w_stream) return NEFII_E_ARG;
    if (nefii_mlp_stream_bytes(h_mlp) == 0) return NEFII_E_UNSUPPORTED;
    const int G = mstream_units(h_mlp);
    hipLaunchKernelGGL(pack_mlp_stream_kernel, dim3(G, 8), dim3(256), 0, (hipStream_t)stream, *h_mlp, (half8 *)w_stream, G);
    HIP_CHECK_LAUNCH();
    const int Gb = mstream_units_bwd(h_mlp);
    if (Gb > 0) {
        hipLaunchKernelGGL(pack_mlp_stream_bwd_kernel, dim3(Gb, 8), dim3(256), 0, (hipStream_t)stream, *h_mlp,
                           (half8 *)w_stream + (size_t)8 * G * 256, Gb);
        HIP_CHECK_LAUNCH();
    }
    return 0;
}

// H16: the stash is what the backward pass and the weight gradients read it as - fp16.  stash_v = [n_layers - 1][n][stash_stride]
// HALVES holding 16 h_l (A16_SCALE: the very hi halves this kernel parks in its own activation image, so the weight-gradient
// GEMM sees bit for bit the operand it used to round from fp32 itself), z_last = [n][8] fp32 pre-activations of the head,
// x0_16 = [n][k_x(layer 0) + EW] halves: layer 0's input image (16 x [features | encodings, zero-padded]) - the B operand of
// layer 0's weight gradient, which otherwise needs nefii_encode_inputs' fp32 matrix rebuilt in the backward pass.
template <int QT, int EW, bool H16>
__global__ __launch_bounds__(512, 2) void mlp_forward16q_kernel(nefii_mlp m, const float *__restrict__ in_a,
                                                               const float *__restrict__ in_b,
                                                               const float *__restrict__ in_c,
                                                               const float *__restrict__ feat, int64_t n,
                                                               float *__restrict__ out, int out_stride,
                                                               float *__restrict__ hidden_out, int hid_stride,
                                                               void *__restrict__ stash_v, int stash_stride,
                                                               float *__restrict__ z_last, _Float16 *__restrict__ x0_16, int G) {
    NEFII_CLAIM_SIMD_2();
    float *const stash = H16 ? nullptr : static_cast<float *>(stash_v);
    _Float16 *const stash16 = H16 ? static_cast<_Float16 *>(stash_v) : nullptr;
    constexpr int ROWS = 16 * QT, XP = LdsM<QT, EW>::XP, EP = 512, NW = 8, NJ = 4 * QT, RT = (QT + 1) / 2;
    __shared__ LdsM<QT, EW> lds;
    __shared__ float raw[ROWS * 9];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int NH = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    const int64_t n_tiles = (n + ROWS - 1) / ROWS;
    {   // the K-padded stream multiplies what follows a layer's own columns by zero weights: keep the image finite
        uint32_t *z = reinterpret_cast<uint32_t *>(&lds);
        for (int i = tid; i < (int)(sizeof(lds) / 4); i += 512) z[i] = 0u;
    }
    P16<8>::Stage b[4];
    PCursor cur;
    cur.bytes = (unsigned)G * 4096;
    cur.base = reinterpret_cast<const half8 *>(m.w_stream) + (size_t)wave * G * 256 + lane;
    cur.off = 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) pload<8>(b[u], cur);
    const int boff = 64 * wave + lane;
    const _Float16 *qh0 = lds.Xh + (lane & 15) * XP + 8 * (lane >> 4), *ql0 = lds.Xl + (lane & 15) * XP + 8 * (lane >> 4);
    const int w0 = enc_width(m.enc_freqs[0]), w1 = enc_width(m.enc_freqs[1]), w2 = enc_width(m.enc_freqs[2]);
    const int F = m.feat_width, kx0 = m.layer[0].k_x;
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * ROWS;
        for (int i = tid; i < ROWS * 9; i += 512) {
            const int p = i / 9, c = i - 9 * p, which = c / 3;
            const float *src = which == 0 ? in_a : (which == 1 ? in_b : in_c);
            int64_t idx = base + p;
            if (idx >= n) idx = n - 1;
            raw[i] = (src && m.enc_freqs[which] >= 0) ? src[idx * 3 + (c - 3 * which)] : 0.f;
        }
        if (feat && F == kx0 && (F & 3) == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0) {
            const int q4 = kx0 >> 2;                    // 16-byte loads, four features per thread and trip
            for (int i = tid; i < ROWS * q4; i += 512) {
                const int p = i / q4, f = 4 * (i - p * q4);
                int64_t idx = base + p;
                if (idx >= n) idx = n - 1;
                const float4v v = *reinterpret_cast<const float4v *>(feat + idx * F + f) * A16_SCALE;
                const half4 hi = __builtin_convertvector(v, half4);
                *reinterpret_cast<half4 *>(lds.Xh + p * XP + EP - kx0 + f) = hi;
                *reinterpret_cast<half4 *>(lds.Xl + p * XP + EP - kx0 + f) =
                    __builtin_convertvector(v - __builtin_convertvector(hi, float4v), half4);
            }
        } else {
            for (int i = tid; i < ROWS * kx0; i += 512) {
                const int p = i / kx0, f = i - p * kx0;
                int64_t idx = base + p;
                if (idx >= n) idx = n - 1;
                const float v = (feat && f < F) ? feat[idx * F + f] : 0.f;
                split16a(v, lds.Xh[p * XP + EP - kx0 + f], lds.Xl[p * XP + EP - kx0 + f]);
            }
        }
        float bnext = m.layer[0].bias[boff];
        __syncthreads();
        for (int i = tid; i < ROWS * EW; i += 512) {
            const int p = i / EW, c = i - p * EW;
            float val = 0.f;
            if (c < w0) val = enc_value(raw + p * 9, c);
            else if (c < w0 + w1) val = enc_value(raw + p * 9 + 3, c - w0);
            else if (c < w0 + w1 + w2) val = enc_value(raw + p * 9 + 6, c - w0 - w1);
            split16a(val, lds.Xh[p * XP + EP + c], lds.Xl[p * XP + EP + c]);
        }
        __syncthreads();
        if (H16 && x0_16) {     // the finished input image's hi halves, 16 bytes per thread and trip
            const int C = (kx0 + EW) >> 3;
            for (int i = tid; i < ROWS * C; i += 512) {
                const int p = i / C, c = i - p * C;
                if (base + p < n)
                    *reinterpret_cast<half8 *>(x0_16 + (size_t)(base + p) * (kx0 + EW) + 8 * c) =
                        *reinterpret_cast<const half8 *>(lds.Xh + p * XP + EP - kx0 + 8 * c);
            }
        }
        for (int l = 0; l < NH; ++l) {
            const nefii_layer &L = m.layer[l];
            const int units = m_units(L);
            const _Float16 *ah = qh0 + (EP - L.k_x), *al = ql0 + (EP - L.k_x);
            const float *bp = m.layer[l + 1 < NH ? l + 1 : l].bias + boff;
            asm volatile("" ::"s"(units), "v"(ah), "v"(al), "v"(bp));
            __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): known state for the waitcnt pass (see "16p")
            __builtin_amdgcn_sched_barrier(0);
            const float bvec = bnext;
            f32x4 acc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
            QAct<QT> a[2];
            qload_a<QT, XP>(a[0], ah, al, 0);
            qgemm<QT, 4, XP>(units, b, a, cur, ah, al, acc);
            bnext = *bp;
            __builtin_amdgcn_sched_barrier(0);
            half4 phi[NJ], plo[NJ];
            const int bsrc = __builtin_bit_cast(int, bvec);
            const bool pre_last = hidden_out != nullptr && l == NH - 1;
            auto epilogue = [&](auto actc) {        // the activation id resolved once per layer, not per value
                constexpr int ACT = decltype(actc)::value;
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    float4v bs;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        bs[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (16 * ft + 4 * (lane >> 4) + k), bsrc));
                    const int f0 = 64 * wave + 16 * ft + 4 * (lane >> 4);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const int j = ft * QT + qt;
                        float4v hv;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float z = __builtin_fmaf(acc[j][k], inv_scale, bs[k]);
                            hv[k] = ACT == NEFII_ACT_ELU ? elu_fast(z) : act_fwd(z, ACT);
                        }
                        const int64_t row = base + 16 * qt + (lane & 15);
                        if (row < n) {
                            if (stash) *reinterpret_cast<float4v *>(stash + ((size_t)l * n + row) * stash_stride + f0) = hv;
                            if (pre_last) {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (f0 + k < L.n_out) hidden_out[(size_t)row * hid_stride + f0 + k] = hv[k];
                            }
                        }
                        const float4v hs = hv * A16_SCALE;
                        const half4 hi = __builtin_convertvector(hs, half4);
                        phi[j] = hi;
                        plo[j] = __builtin_convertvector(hs - __builtin_convertvector(hi, float4v), half4);
                        if (H16 && stash16 && row < n)
                            *reinterpret_cast<half4 *>(stash16 + ((size_t)l * n + row) * stash_stride + f0) = hi;
                    }
                }
            };
            if (m.act == NEFII_ACT_RELU)
                epilogue(std::integral_constant<int, NEFII_ACT_RELU>{});
            else if (m.act == NEFII_ACT_ELU)
                epilogue(std::integral_constant<int, NEFII_ACT_ELU>{});
            else
                epilogue(std::integral_constant<int, NEFII_ACT_SOFTPLUS100>{});
            __syncthreads();
            _Float16 *xh = lds.Xh + (EP - L.n_pad), *xl = lds.Xl + (EP - L.n_pad);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const int f0 = 64 * wave + 16 * ft + 4 * (lane >> 4);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int query = 16 * qt + (lane & 15);
                    *reinterpret_cast<half4 *>(xh + query * XP + f0) = phi[ft * QT + qt];
                    *reinterpret_cast<half4 *>(xl + query * XP + f0) = plo[ft * QT + qt];
                }
            }
            __syncthreads();
        }
        // last layer: one 32-feature tile, 32x32x16 fragments of its own w_f16x3, K split over the waves
        {
            const int r = lane & 31, h = lane >> 5;
            const nefii_layer &L = m.layer[NH];
            const half8 *wl = reinterpret_cast<const half8 *>(L.w_f16x3) + lane;        // NT = 1
            const int ksw = (L.k_x >> 4) / NW;
            f32x16 acc2[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[rt][i] = 0.f;
            for (int u = 0; u < ksw; ++u) {
                const int s = wave * ksw + u;
                const half8 wh = wl[(size_t)s * 128], wlo = wl[(size_t)s * 128 + 64];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int rowc = 32 * rt + r < ROWS ? 32 * rt + r : ROWS - 1;       // 48-row tiles: half a row tile
                    const half8 xh8 = *reinterpret_cast<const half8 *>(lds.Xh + rowc * XP + 8 * h + 16 * s);
                    const half8 xl8 = *reinterpret_cast<const half8 *>(lds.Xl + rowc * XP + 8 * h + 16 * s);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh8, acc2[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xh8, acc2[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl8, acc2[rt], 0, 0, 0);
                }
            }
            __syncthreads();            // every wave is done reading the image: its lo half becomes the reduction scratch
            float *psum = reinterpret_cast<float *>(lds.Xl);       // [wave][row][8 features]
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                if (32 * rt + r < ROWS) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) psum[(wave * ROWS + 32 * rt + r) * 8 + i + 4 * h] = acc2[rt][i];
                }
            __syncthreads();
            for (int i = tid; i < ROWS * 8; i += 512) {
                const int p = i >> 3, c = i & 7;
                if (c < L.n_out && base + p < n) {
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sum += psum[(w * ROWS + p) * 8 + c];
                    const float z = sum * inv_scale + L.bias[c];
                    if (stash) stash[((size_t)NH * n + base + p) * stash_stride + c] = z;
                    if (H16 && z_last) z_last[(size_t)(base + p) * 8 + c] = z;
                    out[(size_t)(base + p) * out_stride + c] = head_fwd(z, m.head);
                }
            }
            __syncthreads();
            // The scratch held fp32 partial sums - NW * ROWS * 8 of them, 16 KiB: rows 0-13 of the lo image.  What the next
            // tile's zero-weight k-steps read there must be finite halves (0 x NaN = NaN): the row's 8 pad columns, which
            // no loader rewrites, and - nets whose layer 0 takes fewer than 512 features - the next row's first columns.
            // All of it is cleared (until round 3 only the first half was), and the barrier keeps a late wave's zeros from
            // landing on lo halves the next tile's feature staging has already written.
            for (int i = tid; i < NW * ROWS * 8; i += 512) reinterpret_cast<uint32_t *>(lds.Xl)[i] = 0u;
            __syncthreads();
        }
    }
}

static int check_mlp16(const nefii_mlp *m, bool bwd) {
    int rc = check_mlp(m, false);
    if (rc) return rc;
    for (int l = 0; l < m->n_layers; ++l) {
        if (!m->layer[l].w_f16x3) return NEFII_E_ARG;
        if (bwd && l > 0 && !m->layer[l].w_bwd_f16x3) return NEFII_E_ARG;
        if ((m->layer[l].k_x | m->layer[l].k_e | m->layer[l].n_pad) & 15) return NEFII_E_SHAPE;
    }
    return 0;
}

// NEFII_MLP_STREAM=0: keep the 32-row kernel (A/B measurements)
static bool mlp_stream_enabled() {
    static const bool v = [] {
        const char *e = getenv("NEFII_MLP_STREAM");
        return !(e && atoi(e) == 0);
    }();
    return v;
}

extern "C" int nefii_mlp_forward_f16(const nefii_mlp *h_mlp, const float *in_a, const float *in_b, const float *in_c,
                                     const float *feat, int64_t n, float *out, int out_stride, float *hidden_out,
                                     int hid_stride, float *stash, int stash_stride, int single_pass, void *stream) {
    int rc = check_mlp16(h_mlp, false);
    if (rc) return rc;
    if (n <= 0) return 0;
    if (!out) return NEFII_E_ARG;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    if (!single_pass && h_mlp->w_stream && mlp_stream_enabled()) {
        const int ew = mstream_shape(h_mlp), G = mstream_units(h_mlp);
        if (ew == 64) {
            const int64_t t = (n + 63) / 64;
            hipLaunchKernelGGL((mlp_forward16q_kernel<4, 64, false>), dim3((int)(t < 256 ? t : 256)), dim3(512), 0,
                               (hipStream_t)stream, *h_mlp, in_a, in_b, in_c, feat, n, out, out_stride, hidden_out, hid_stride,
                               (void *)stash, stash_stride, (float *)nullptr, (_Float16 *)nullptr, G);
            HIP_CHECK_LAUNCH();
            return 0;
        }
        if (ew == 96) {
            const int64_t t = (n + 63) / 64;
            hipLaunchKernelGGL((mlp_forward16q_kernel<4, 96, false>), dim3((int)(t < 256 ? t : 256)), dim3(512), 0,
                               (hipStream_t)stream, *h_mlp, in_a, in_b, in_c, feat, n, out, out_stride, hidden_out, hid_stride,
                               (void *)stash, stash_stride, (float *)nullptr, (_Float16 *)nullptr, G);
            HIP_CHECK_LAUNCH();
            return 0;
        }
    }
    if (single_pass)
        hipLaunchKernelGGL(mlp_forward16_kernel<true>, dim3(grid_for(n_tiles, 1)), dim3(WG), 0, (hipStream_t)stream, *h_mlp,
                           in_a, in_b, in_c, feat, n, out, out_stride, hidden_out, hid_stride, stash, stash_stride);
    else
        hipLaunchKernelGGL(mlp_forward16_kernel<false>, dim3(grid_for(n_tiles, 1)), dim3(WG), 0, (hipStream_t)stream, *h_mlp,
                           in_a, in_b, in_c, feat, n, out, out_stride, hidden_out, hid_stride, stash, stash_stride);
    HIP_CHECK_LAUNCH();
    return 0;
}

// ---- the fp16 training path of the streamed nets (ABI 10): stash and dz in halves -------------------------------------------
// What the backward pass and the weight-gradient GEMMs consume is fp16 anyway: the stash's activations as the GEMM's B
// operand and (through act') as the backward epilogue's factor, dz as the GEMM's A operand and the next layer's image.  Kept in
// fp32 they were written once and read twice at twice the bytes (1.4-2.3 GB per call on config 3).  nefii_mlp_h16_supported:
// the net runs on the streamed kernels forward AND backward (512-wide hidden layers, no skip layer, head of <= 8 outputs).
extern "C" int nefii_mlp_h16_supported(const nefii_mlp *h_mlp) {
    if (!h_mlp || check_mlp16(h_mlp, true)) return 0;
    return h_mlp->w_stream && mlp_stream_enabled() && mstream_shape(h_mlp) && mstream_units_bwd(h_mlp) > 0;
}

// width of nefii_mlp_forward_f16h's x0_16 rows: layer 0's padded feature columns + its 64 or 96 encoding columns (0: no h16 path)
extern "C" int nefii_mlp_x0_width(const nefii_mlp *h_mlp) {
    if (!nefii_mlp_h16_supported(h_mlp)) return 0;
    return h_mlp->layer[0].k_x + mstream_shape(h_mlp);
}

extern "C" int nefii_mlp_forward_f16h(const nefii_mlp *h_mlp, const float *in_a, const float *in_b, const float *in_c,
                                      const float *feat, int64_t n, float *out, int out_stride, float *hidden_out,
                                      int hid_stride, void *stash16, int stash_stride, float *z_last, void *x0_16,
                                      void *stream) {
    int rc = check_mlp16(h_mlp, false);
    if (rc) return rc;
    if (!nefii_mlp_h16_supported(h_mlp)) return NEFII_E_SHAPE;
    if (n <= 0) return 0;
    if (!out || !stash16 || !z_last || (stash_stride & 3)) return NEFII_E_ARG;
    const int ew = mstream_shape(h_mlp), G = mstream_units(h_mlp);
    const int64_t t = (n + 63) / 64;
    if (ew == 64)
        hipLaunchKernelGGL((mlp_forward16q_kernel<4, 64, true>), dim3((int)(t < 256 ? t : 256)), dim3(512), 0, (hipStream_t)stream,
                           *h_mlp, in_a, in_b, in_c, feat, n, out, out_stride, hidden_out, hid_stride, stash16, stash_stride,
                           z_last, (_Float16 *)x0_16, G);
    else
        hipLaunchKernelGGL((mlp_forward16q_kernel<4, 96, true>), dim3((int)(t < 256 ? t : 256)), dim3(512), 0, (hipStream_t)stream,
                           *h_mlp, in_a, in_b, in_c, feat, n, out, out_stride, hidden_out, hid_stride, stash16, stash_stride,
                           z_last, (_Float16 *)x0_16, G);
    HIP_CHECK_LAUNCH();
    return 0;
}

// S = 2^(8 - ceil(log2(max |d_out|))): the largest gradient entering the backward GEMMs becomes ~256 in fp16, with a
// factor ~256 of headroom for the layers to amplify it and ~2^-32 of range below (1 when d_out is all zero).
__global__ void grad_scale_kernel(const float *__restrict__ d, int64_t count, float *__restrict__ scale) {
    __shared__ float wmax[16];
    float mx = 0.f;
    for (int64_t i = threadIdx.x; i < count; i += blockDim.x) {
        const float v = fabsf(d[i]);
        mx = (v < 3.0e38f && v > mx) ? v : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mx = fmaxf(mx, wmax[w]);
        float s = 1.f;
        if (mx > 0.f) {
            int e;
            frexpf(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)
            e = 8 - e;
            e = e < -40 ? -40 : (e > 60 ? 60 : e);
            s = ldexpf(1.f, e);
        }
        scale[0] = s;
    }
}

extern "C" int nefii_mlp_grad_scale(const float *d_out, int64_t count, float *scale, void *stream) {
    if (!d_out || !scale || count < 0) return NEFII_E_ARG;
    hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_out, count, scale);
    HIP_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256, 2) void mlp_backward16_kernel(nefii_mlp m, const float *__restrict__ d_out,
                                                                int out_stride, const float *__restrict__ stash,
                                                                int stash_stride, int64_t n, float *__restrict__ dz,
                                                                int dz_stride, const float *__restrict__ scale) {
    __shared__ _Float16 Xh[TILE * XS16];
    const int tid = threadIdx.x;
    const int Lm1 = m.n_layers - 1;
    const float S = scale[0];
    const float inv = 1.f / (W16_SCALE * S);
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * TILE;
        {   // seed: dz_{L-1} = d_out * head'(pre)
            const nefii_layer &L = m.layer[Lm1];
            for (int i = tid; i < TILE * L.n_pad; i += WG) {
                const int p = i / L.n_pad, c = i - p * L.n_pad;
                float v = 0.f;
                if (base + p < n && c < L.n_out) {
                    const float pre = stash[((size_t)Lm1 * n + base + p) * stash_stride + c];
                    const float y = head_fwd(pre, m.head);
                    v = d_out[(size_t)(base + p) * out_stride + c] * head_bwd_from_out(y, pre, m.head);
                }
                Xh[p * XS16 + c] = (_Float16)(v * S);
                if (base + p < n) dz[((size_t)Lm1 * n + base + p) * dz_stride + c] = v;
            }
        }
        __syncthreads();
        for (int l = Lm1; l >= 1; --l) {
            const nefii_layer &L = m.layer[l];
            // dH_{l-1}[32 x k_x] = dZ_l[32 x n_pad] * W_l   (only the hidden block of the inputs)
            f32x16 acc[4];
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            const int NTs = (L.k_x + L.k_e) >> 5, ntc = L.k_x >> 5;
            int ntw = (ntc - wave + 3) >> 2;
            if (ntw < 0) ntw = 0;
            zero_acc(acc);
            gemm_block16<true>(Xh, Xh, XS16, L.n_pad >> 4, reinterpret_cast<const half8 *>(L.w_bwd_f16x3), NTs, wave, lane,
                               ntw, acc);
            __syncthreads();
            NEFII_ACT_SWITCH(m.act, {
                NEFII_FOR_ACC(acc, ntw, {
                    const bool live = (base + row) < n;
                    float v = 0.f;
                    if (live) {
                        const float hprev = stash[((size_t)(l - 1) * n + base + row) * stash_stride + col];
                        v = val * inv * act_bwd_from_out(hprev, ACT);
                        dz[((size_t)(l - 1) * n + base + row) * dz_stride + col] = v;
                    }
                    Xh[row * XS16 + col] = (_Float16)(v * S);
                })
            })
            __syncthreads();
        }
    }
}

// The one-pass fp16 backward on the fragment stream (nets mstream_shape() takes): the single-pass evaluator's machine
// (mlp_tile.h "16s": hi fragments, one 32-deep k-step of the wave's four feature tiles per 4 KiB unit, hi-only activation
// image of 64 rows) over the transposed layers, last to first.  The image holds S dz_l in fp16, a layer's k-loop contracts
// over its outputs, the epilogue multiplies by act'(h_{l-1}) from the forward's stash, writes dz_{l-1} (fp32, row-major:
// what nefii_mlp_wgrad_f16 reads) and parks S dz_{l-1} as the next image.  The 32-row kernel above fetched each layer's
// fragments per tile with nothing in flight (~30 GB/s per CU).
// H16 (nefii_mlp_backward_f16h): stash_v = the forward's fp16 stash (16 h_l), z_last its head pre-activations, dz_v =
// [n_layers][n][dz_stride] HALVES holding S dz_l - the values this kernel parks as the next layer's image anyway.
template <bool H16>
__global__ __launch_bounds__(512, 2) void mlp_backward16s_kernel(nefii_mlp m, const float *__restrict__ d_out, int out_stride,
                                                                const void *__restrict__ stash_v, int stash_stride,
                                                                const float *__restrict__ z_last, int64_t n,
                                                                void *__restrict__ dz_v, int dz_stride,
                                                                const float *__restrict__ scale, size_t stream_off, int G) {
    NEFII_CLAIM_SIMD_2();
    const float *const stash = static_cast<const float *>(stash_v);
    const _Float16 *const stash16 = static_cast<const _Float16 *>(stash_v);
    float *const dz = static_cast<float *>(dz_v);
    _Float16 *const dz16 = static_cast<_Float16 *>(dz_v);
    constexpr int QT = 4, FT = 4, ROWS = 64, XP = QGeo<4>::XP, NJ = FT * QT;
    __shared__ LdsS<4, ROWS> lds;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int Lm1 = m.n_layers - 1;
    const float S = scale[0];
    const float inv = 1.f / (W16_SCALE * S);
    const int64_t n_tiles = (n + ROWS - 1) / ROWS;
    {   // the K-padded stream meets whatever follows a layer's own columns with zero weights: keep the image finite
        uint32_t *z = reinterpret_cast<uint32_t *>(&lds);
        for (int i = tid; i < (int)(sizeof(lds) / 4); i += 512) z[i] = 0u;
    }
    SStage<4> b[4];
    PCursor cur;
    cur.bytes = (unsigned)G * 4096;
    cur.base = reinterpret_cast<const half8 *>(m.w_stream) + stream_off + (size_t)wave * G * 256 + lane;
    cur.off = 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) sload<4>(b[u], cur);
    const _Float16 *qh0 = lds.Xh + (lane & 15) * XP + 8 * (lane >> 4);
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * ROWS;
        {   // seed: dz_last = d_out * head'(pre); the last layer's n_pad columns of the image (zeros past n_out)
            const nefii_layer &L = m.layer[Lm1];
            const int kpad = mb_units(L) * 32;          // the columns the first k-loop reads: zeros past the layer's outputs
            for (int i = tid; i < ROWS * kpad; i += 512) {
                const int p = i / kpad, c = i - p * kpad;
                float v = 0.f;
                if (base + p < n && c < L.n_pad) {
                    if (c < L.n_out) {
                        const float pre = H16 ? z_last[(size_t)(base + p) * 8 + c]
                                              : stash[((size_t)Lm1 * n + base + p) * stash_stride + c];
                        const float y = head_fwd(pre, m.head);
                        v = d_out[(size_t)(base + p) * out_stride + c] * head_bwd_from_out(y, pre, m.head);
                    }
                    if (H16) dz16[((size_t)Lm1 * n + base + p) * dz_stride + c] = (_Float16)(v * S);
                    else dz[((size_t)Lm1 * n + base + p) * dz_stride + c] = v;
                }
                lds.Xh[p * XP + c] = (_Float16)(v * S);
            }
        }
        __syncthreads();
        for (int l = Lm1; l >= 1; --l) {
            const nefii_layer &L = m.layer[l];
            const int units = mb_units(L);
            asm volatile("" ::"s"(units));
            __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): known state for the waitcnt pass (see "16p")
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
            SAct<QT> a[2];
            sload_a<QT, XP>(a[0], qh0, 0);
            sgemm<QT, FT, true>(units, b, a, cur, qh0, acc);
            // dH_{l-1} -> dz_{l-1} = dH * act'(h_{l-1}); rows past n carry zeros
            half4 phi[NJ];
            const float *sp = stash + (size_t)(l - 1) * n * stash_stride;
            float *dp = dz + (size_t)(l - 1) * n * dz_stride;
            const _Float16 *sp16 = stash16 + (size_t)(l - 1) * n * stash_stride;
            _Float16 *dp16 = dz16 + (size_t)(l - 1) * n * dz_stride;
            auto epilogue = [&](auto actc) {
                constexpr int ACT = decltype(actc)::value;
#pragma unroll
                for (int ft = 0; ft < FT; ++ft) {
                    const int f0 = 64 * wave + 16 * ft + 4 * (lane >> 4);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const int j = ft * QT + qt;
                        const int64_t row = base + 16 * qt + (lane & 15);
                        float4v v = {0.f, 0.f, 0.f, 0.f};
                        if (row < n) {
                            float4v h;
                            if (H16)
                                h = __builtin_convertvector(*reinterpret_cast<const half4 *>(sp16 + (size_t)row * stash_stride + f0),
                                                            float4v) * (1.f / A16_SCALE);
                            else
                                h = *reinterpret_cast<const float4v *>(sp + (size_t)row * stash_stride + f0);
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = acc[j][k] * inv * act_bwd_from_out(h[k], ACT);
                            if (!H16) *reinterpret_cast<float4v *>(dp + (size_t)row * dz_stride + f0) = v;
                        }
                        phi[j] = __builtin_convertvector(v * S, half4);
                        if (H16 && row < n) *reinterpret_cast<half4 *>(dp16 + (size_t)row * dz_stride + f0) = phi[j];
                    }
                }
            };
            if (m.act == NEFII_ACT_RELU)
                epilogue(std::integral_constant<int, NEFII_ACT_RELU>{});
            else if (m.act == NEFII_ACT_ELU)
                epilogue(std::integral_constant<int, NEFII_ACT_ELU>{});
            else
                epilogue(std::integral_constant<int, NEFII_ACT_SOFTPLUS100>{});
            __syncthreads();
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                const int f0 = 64 * wave + 16 * ft + 4 * (lane >> 4);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    *reinterpret_cast<half4 *>(lds.Xh + (16 * qt + (lane & 15)) * XP + f0) = phi[ft * QT + qt];
            }
            __syncthreads();
        }
    }
}

extern "C" int nefii_mlp_backward_f16(const nefii_mlp *h_mlp, const float *d_out, int out_stride, const float *stash,
                                      int stash_stride, int64_t n, float *dz, int dz_stride, const float *scale,
                                      void *stream) {
    int rc = check_mlp16(h_mlp, true);
    if (rc) return rc;
    if (n <= 0) return 0;
    if (!d_out || !stash || !dz || !scale) return NEFII_E_ARG;
    if (h_mlp->w_stream && mlp_stream_enabled() && mstream_shape(h_mlp) && mstream_units_bwd(h_mlp) > 0 &&
        (stash_stride & 3) == 0 && (dz_stride & 3) == 0) {
        const int64_t t = (n + 63) / 64;
        hipLaunchKernelGGL(mlp_backward16s_kernel<false>, dim3((int)(t < 256 ? t : 256)), dim3(512), 0, (hipStream_t)stream,
                           *h_mlp, d_out, out_stride, (const void *)stash, stash_stride, (const float *)nullptr, n, (void *)dz,
                           dz_stride, scale, (size_t)8 * mstream_units(h_mlp) * 256, mstream_units_bwd(h_mlp));
        HIP_CHECK_LAUNCH();
        return 0;
    }
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    hipLaunchKernelGGL(mlp_backward16_kernel, dim3(grid_for(n_tiles, 2)), dim3(WG), 0, (hipStream_t)stream, *h_mlp, d_out,
                       out_stride, stash, stash_stride, n, dz, dz_stride, scale);
    HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int nefii_mlp_backward_f16h(const nefii_mlp *h_mlp, const float *d_out, int out_stride, const void *stash16,
                                       int stash_stride, const float *z_last, int64_t n, void *dz16, int dz_stride,
                                       const float *scale, void *stream) {
    int rc = check_mlp16(h_mlp, true);
    if (rc) return rc;
    if (!nefii_mlp_h16_supported(h_mlp)) return NEFII_E_SHAPE;
    if (n <= 0) return 0;
    if (!d_out || !stash16 || !z_last || !dz16 || !scale || (stash_stride & 3) || (dz_stride & 3)) return NEFII_E_ARG;
    const int64_t t = (n + 63) / 64;
    hipLaunchKernelGGL(mlp_backward16s_kernel<true>, dim3((int)(t < 256 ? t : 256)), dim3(512), 0, (hipStream_t)stream, *h_mlp,
                       d_out, out_stride, stash16, stash_stride, z_last, n, dz16, dz_stride, scale,
                       (size_t)8 * mstream_units(h_mlp) * 256, mstream_units_bwd(h_mlp));
    HIP_CHECK_LAUNCH();
    return 0;
}

// dW[n][k] = scale * sum_p dz[p][n] * x[p][k] on v_mfma_f32_32x32x16_f16: 16 points per instruction.  A = (S dz)^T
// fragment (lane (n, p-group): 8 consecutive points), B = x fragment (lane (k, p-group)); fp32 loads (coalesced over the
// 32 lanes of a point row), converted on the fly.  Same block shape / point split / atomics as mlp_wgrad_kernel.
// DZ16 / X16: operands stored as halves (S dz / 16 h), see mlp_wgrad16t_kernel.
// (bx, by, bz, gz: the block's place in the layer's own grid - the batched launch below maps one flat grid onto several layers)
template <bool DZ16, bool X16>
__device__ __forceinline__ void wgrad16_body(const void *__restrict__ dz_v, int dz_stride, const void *__restrict__ x_v,
                                             int x_stride, int64_t P, int n_out, int k_in, float scale,
                                             const float *__restrict__ gscale, float *__restrict__ dW, float *__restrict__ db,
                                             int atomic, int bx, int by, int bz, int gz) {
    const float *const dz = static_cast<const float *>(dz_v), *const x = static_cast<const float *>(x_v);
    const _Float16 *const dz16 = static_cast<const _Float16 *>(dz_v), *const x16 = static_cast<const _Float16 *>(x_v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int n0 = bx * 64 + (wave & 1) * 32;
    const int k0 = by * 256 + (wave >> 1) * 128;
    const int64_t chunk = ((P + gz - 1) / gz + 15) & ~(int64_t)15;
    const int64_t p_begin = (int64_t)bz * chunk;
    const int64_t p_end = p_begin + chunk < P ? p_begin + chunk : P;
    const float S = gscale[0];
    f32x16 acc[4];
    zero_acc(acc);
    const bool n_ok = (n0 + i) < n_out;
    bool k_ok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) k_ok[t] = (k0 + 32 * t + i) < k_in;
    float bsum = 0.f;
    if (n0 < n_out && k0 < k_in) {
        // every load is unconditional (clamped address, value selected afterwards): a load behind a runtime condition
        // makes hipcc branch around it and wait vmcnt(0) per element - 40 dependent round trips per batch
        const int nc = n_ok ? n0 + i : n_out - 1;
        int kc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) kc[t] = k_ok[t] ? k0 + 32 * t + i : k_in - 1;
        const int64_t p_last = p_end - 1;
        constexpr int U = 2;                       // 16-point batches in flight: all loads of U batches, then their MFMAs
        for (int64_t pb = p_begin; pb < p_end; pb += 16 * U) {
            float a[U][8], b[U][4][8];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t p = pb + 16 * u + 8 * h + j;
                    const int64_t pc = p < p_last ? p : p_last;
                    a[u][j] = DZ16 ? (float)dz16[pc * dz_stride + nc] : dz[pc * dz_stride + nc];
#pragma unroll
                    for (int t = 0; t < 4; ++t) b[u][t][j] = X16 ? (float)x16[pc * x_stride + kc[t]] : x[pc * x_stride + kc[t]];
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                half8 af;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = (pb + 16 * u + 8 * h + j) < p_end && n_ok;
                    const float av = ok ? a[u][j] : 0.f;
                    bsum += av;
                    af[j] = DZ16 ? (_Float16)av : (_Float16)(av * S);       // (halves come back exactly)
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    half8 bf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bf[j] = (_Float16)(k_ok[t] ? b[u][t][j] : 0.f);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[t], 0, 0, 0);
                }
            }
        }
        const float os = scale / S * (X16 ? 1.f / A16_SCALE : 1.f);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * h, k = k0 + 32 * t + i;
                if (nn < n_out && k < k_in) {
                    const float v = acc[t][r] * os;
                    if (atomic) atomicAdd(&dW[(size_t)nn * k_in + k], v);
                    else dW[(size_t)nn * k_in + k] = v;
                }
            }
    }
    if (db && by == 0 && (wave >> 1) == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (DZ16) bsum /= S;
        if (h == 0 && n_ok) {
            if (atomic) atomicAdd(&db[n0 + i], bsum);
            else db[n0 + i] = bsum;
        }
    }
}

template <bool DZ16, bool X16>
__global__ __launch_bounds__(256) void mlp_wgrad16_kernel(const void *__restrict__ dz_v, int dz_stride,
                                                          const void *__restrict__ x_v, int x_stride, int64_t P, int n_out,
                                                          int k_in, float scale, const float *__restrict__ gscale,
                                                          float *__restrict__ dW, float *__restrict__ db, int atomic) {
    wgrad16_body<DZ16, X16>(dz_v, dz_stride, x_v, x_stride, P, n_out, k_in, scale, gscale, dW, db, atomic, blockIdx.x, blockIdx.y,
                            blockIdx.z, gridDim.z);
}

// ---- the same product as a blocked GEMM (layers of at least 64 x 64 weights) ---------------------------------------------
// The kernel above feeds each 32x32x16 MFMA from 40 scalar loads per lane (dz and x are POINT-major, the MFMA wants 8
// consecutive points per lane): it is bound by vector-memory issue.  Here a workgroup of 4 waves owns a 128 x 128 block of
// dW over a slice of the points: 32-point slabs of dz (x S) and x go through LDS as fp16 [point][column] rows (16-byte
// global loads, coalesced 512 B per row; two slabs in flight), and the MFMA operands come back TRANSPOSED through
// ds_read_b64_tr_b16 (gfx950: a 16-lane group reads 4 rows x 16 columns and each lane receives one column) - two reads
// per operand fragment, no shuffles.  Row stride 160 halves (320 B): the four rows of a block fall 16 banks apart and the
// half-wave's second block 8 banks further, so every transposed read is conflict-free.  Blocks of one point slice are
// mapped to ONE XCD (workgroups are dealt round-robin over the 8 XCDs): its 16 column blocks share the slice's rows in
// that XCD's L2 instead of fetching them eight times.  Partial products leave with one atomic per element and block.
constexpr int WG2_ROWS = 32, WG2_STRIDE = 160, WG2_POINTS = 4096;
typedef short short4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ half8 wg2_fragment(const _Float16 *tile, int col0, int row0, int lane) {
    // block rows row0 .. row0+3 (then +4 .. +7), columns col0 + 16 ((lane>>4)&1) + 0..15; lane 4q+p of a 16-lane group
    // addresses row q, columns 4p .. 4p+3
    const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3;
    const _Float16 *a = tile + (row0 + 8 * (lane >> 5) + q) * WG2_STRIDE + col0 + 16 * ((lane >> 4) & 1) + 4 * pp;
    typedef short4v __attribute__((address_space(3))) * lds_ptr;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * WG2_STRIDE));
    const half4 l4 = __builtin_bit_cast(half4, lo), h4 = __builtin_bit_cast(half4, hi);
    return half8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}

// DZ16 / X16 (nefii_mlp_wgrad_f16h): the operand arrives as the fp16 image itself - dz_v = S dz in halves (the backward
// pass's own rounding), x_v = 16 h in halves (the forward's) - and is copied to LDS with 16-byte loads, eight columns per
// thread, instead of being loaded as fp32 and rounded here: the same operands bit for bit at half the bytes.
template <bool XVEC, bool DZ16, bool X16>
__device__ __forceinline__ void wgrad16t_body(const void *__restrict__ dz_v, int dz_stride, const void *__restrict__ x_v,
                                              int x_stride, int64_t P, int n_out, int k_in, float scale,
                                              const float *__restrict__ gscale, float *__restrict__ dW, float *__restrict__ db,
                                              int splits, int tiles_n, int tiles_k, unsigned bid) {
    __shared__ __attribute__((aligned(16))) _Float16 A[2][WG2_ROWS * WG2_STRIDE], B[2][WG2_ROWS * WG2_STRIDE];
    const float *const dz = static_cast<const float *>(dz_v), *const x = static_cast<const float *>(x_v);
    const _Float16 *const dz16 = static_cast<const _Float16 *>(dz_v), *const x16 = static_cast<const _Float16 *>(x_v);
    // XCD-aware order: the tiles_n * tiles_k blocks of one point slice get consecutive positions on one XCD
    // (bid: the block's index within this layer's own blocks - a multiple of 8 of them, so bid & 7 is the XCD in a batch too)
    const int nt = tiles_n * tiles_k;
    const int xcd = bid & 7, j = bid >> 3;
    const int tile = j % nt, slice = 8 * (j / nt) + xcd;
    if (slice >= splits) return;
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int n0 = 128 * tn, k0 = 128 * tk;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t chunk = ((P + splits - 1) / splits + WG2_ROWS - 1) / WG2_ROWS * WG2_ROWS;
    const int64_t p_begin = (int64_t)slice * chunk;
    const int64_t p_end = p_begin + chunk < P ? p_begin + chunk : P;
    if (p_begin >= p_end) return;
    const float S = gscale[0];
    // staging, fp32 source: thread -> columns 4 c4 .. 4 c4+3 of rows r8, r8 + 8, r8 + 16, r8 + 24
    //          fp16 source: thread -> columns 8 c8 .. 8 c8+7 of rows r16, r16 + 16
    const int c4 = tid & 31, r8 = tid >> 5;
    const int c8 = tid & 15, r16 = tid >> 4;
    const int na = n0 + (DZ16 ? 8 * c8 : 4 * c4), ka = k0 + (X16 ? 8 * c8 : 4 * c4);
    const bool n_in = na + (DZ16 ? 7 : 3) < n_out, k_in_ok = ka + (X16 ? 7 : 3) < k_in;   // whole vector inside (else element-wise)
    float4v ra[4], rb[4];
    half8 ha[2], hb[2];
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const half8 hzero = {0, 0, 0, 0, 0, 0, 0, 0};
    auto fetch = [&](int64_t pb) {
        if (DZ16) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t pr = pb + r16 + 16 * u;
                const int64_t pc = pr < p_end ? pr : p_end - 1;
                half8 a = hzero;
                if (n_in) {
                    a = *reinterpret_cast<const half8 *>(dz16 + pc * dz_stride + na);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] = dz16[pc * dz_stride + (na + e < n_out ? na + e : n_out - 1)];
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] = na + e < n_out ? a[e] : (_Float16)0;
                }
                ha[u] = pr < p_end ? a : hzero;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t pr = pb + r8 + 8 * u;
                const int64_t pc = pr < p_end ? pr : p_end - 1;
                float4v a = {0.f, 0.f, 0.f, 0.f};
                if (n_in) {
                    a = *reinterpret_cast<const float4v *>(dz + pc * dz_stride + na);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = dz[pc * dz_stride + (na + e < n_out ? na + e : n_out - 1)];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = na + e < n_out ? a[e] : 0.f;
                }
                ra[u] = pr < p_end ? a : float4v{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (X16) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t pr = pb + r16 + 16 * u;
                const int64_t pc = pr < p_end ? pr : p_end - 1;
                half8 b = hzero;
                if (XVEC && k_in_ok) {
                    b = *reinterpret_cast<const half8 *>(x16 + pc * x_stride + ka);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) b[e] = x16[pc * x_stride + (ka + e < k_in ? ka + e : k_in - 1)];
#pragma unroll
                    for (int e = 0; e < 8; ++e) b[e] = ka + e < k_in ? b[e] : (_Float16)0;
                }
                hb[u] = pr < p_end ? b : hzero;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t pr = pb + r8 + 8 * u;
                const int64_t pc = pr < p_end ? pr : p_end - 1;
                float4v b = {0.f, 0.f, 0.f, 0.f};
                if (XVEC && k_in_ok) {
                    b = *reinterpret_cast<const float4v *>(x + pc * x_stride + ka);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = x[pc * x_stride + (ka + e < k_in ? ka + e : k_in - 1)];
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = ka + e < k_in ? b[e] : 0.f;
                }
                rb[u] = pr < p_end ? b : float4v{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto stage = [&](int buf) {
        if (DZ16) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += (float)ha[u][e];
                *reinterpret_cast<half8 *>(&A[buf][(r16 + 16 * u) * WG2_STRIDE + 8 * c8]) = ha[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum[e] += ra[u][e];
                *reinterpret_cast<half4 *>(&A[buf][(r8 + 8 * u) * WG2_STRIDE + 4 * c4]) = __builtin_convertvector(ra[u] * S, half4);
            }
        }
        if (X16) {
#pragma unroll
            for (int u = 0; u < 2; ++u) *reinterpret_cast<half8 *>(&B[buf][(r16 + 16 * u) * WG2_STRIDE + 8 * c8]) = hb[u];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                *reinterpret_cast<half4 *>(&B[buf][(r8 + 8 * u) * WG2_STRIDE + 4 * c4]) = __builtin_convertvector(rb[u], half4);
        }
    };
    f32x16 acc[4];
    zero_acc(acc);
    const int an = 64 * (wave & 1), bk = 64 * (wave >> 1);
    fetch(p_begin);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int64_t pb = p_begin; pb < p_end; pb += WG2_ROWS) {
        const bool more = pb + WG2_ROWS < p_end;
        if (more) fetch(pb + WG2_ROWS);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = wg2_fragment(A[buf], an + 32 * t, 16 * ks, lane);
                bf[t] = wg2_fragment(B[buf], bk + 32 * t, 16 * ks, lane);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 2; ++v)
                    acc[2 * t + v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], bf[v], acc[2 * t + v], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const float os = scale / S * (X16 ? 1.f / A16_SCALE : 1.f);
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + an + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h, k = k0 + bk + 32 * v + i;
                if (nn < n_out && k < k_in) atomicAdd(&dW[(size_t)nn * k_in + k], acc[2 * t + v][r] * os);
            }
    if (db && tk == 0) {
        // column sums of this thread's rows: reduce the row groups through LDS, then one atomic per column and block
        __syncthreads();
        float *red = reinterpret_cast<float *>(&A[0][0]);         // [8][128] (fp32 source) / [16][128] (fp16 source)
        if (DZ16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[r16 * 128 + 8 * c8 + e] = bsum[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[r8 * 128 + 4 * c4 + e] = bsum[e];
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < n_out) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < (DZ16 ? 16 : 8); ++g) sum += red[g * 128 + tid];
            atomicAdd(&db[n0 + tid], DZ16 ? sum / S : sum);
        }
    }
}

template <bool XVEC, bool DZ16, bool X16>
__global__ __launch_bounds__(256, 2) void mlp_wgrad16t_kernel(const void *__restrict__ dz_v, int dz_stride,
                                                              const void *__restrict__ x_v, int x_stride, int64_t P, int n_out,
                                                              int k_in, float scale, const float *__restrict__ gscale,
                                                              float *__restrict__ dW, float *__restrict__ db, int splits,
                                                              int tiles_n, int tiles_k) {
    wgrad16t_body<XVEC, DZ16, X16>(dz_v, dz_stride, x_v, x_stride, P, n_out, k_in, scale, gscale, dW, db, splits, tiles_n, tiles_k,
                                   blockIdx.x);
}

// ---- every layer of a net in one launch (nefii_mlp_wgrad_f16h_batch) -----------------------------------------------------
// The weight gradients of a net's layers are independent GEMMs over the same points; one launch per layer (plus one zero-fill
// each) is 2 L dependent launches in the step's tail - config 1's five 18-us launches ran one after the other.  Here one
// grid covers all layers of a group (same kernel, same operand forms): block -> (layer, block of that layer's own grid).
struct WgradBatch {
    nefii_wgrad_item it[NEFII_MAX_WGRAD_ITEMS];
    unsigned first[NEFII_MAX_WGRAD_ITEMS + 1];      // first flat block of item i; first[n] = total
    int aux0[NEFII_MAX_WGRAD_ITEMS], aux1[NEFII_MAX_WGRAD_ITEMS], aux2[NEFII_MAX_WGRAD_ITEMS];
    int n;
};
__device__ __forceinline__ int wgrad_batch_item(const WgradBatch &b, unsigned bid) {
    int i = 0;
    while (i + 1 < b.n && bid >= b.first[i + 1]) ++i;
    return __builtin_amdgcn_readfirstlane(i);
}
// aux0 / aux1 / aux2 = splits / tiles_n / tiles_k
template <bool XVEC, bool X16>
__global__ __launch_bounds__(256, 2) void mlp_wgrad16t_batch_kernel(WgradBatch b, int64_t P, const float *__restrict__ gscale) {
    const int i = wgrad_batch_item(b, blockIdx.x);
    const nefii_wgrad_item &w = b.it[i];
    wgrad16t_body<XVEC, true, X16>(w.dz16, w.dz_stride, w.x, w.x_stride, P, w.n_out, w.k_in, w.scale, gscale, w.dW, w.db, b.aux0[i],
                                   b.aux1[i], b.aux2[i], blockIdx.x - b.first[i]);
}
// aux0 / aux1 / aux2 = grid x / grid y / point splits (grid z)
template <bool X16>
__global__ __launch_bounds__(256) void mlp_wgrad16_batch_kernel(WgradBatch b, int64_t P, const float *__restrict__ gscale) {
    const int i = wgrad_batch_item(b, blockIdx.x);
    const nefii_wgrad_item &w = b.it[i];
    const int local = blockIdx.x - b.first[i], gx = b.aux0[i], gy = b.aux1[i], gz = b.aux2[i];
    wgrad16_body<true, X16>(w.dz16, w.dz_stride, w.x, w.x_stride, P, w.n_out, w.k_in, w.scale, gscale, w.dW, w.db, gz > 1 ? 1 : 0,
                            local % gx, (local / gx) % gy, local / (gx * gy), gz);
}
__global__ void zero_fill_batch_kernel(WgradBatch b) {
    const int i = wgrad_batch_item(b, blockIdx.x);
    const nefii_wgrad_item &w = b.it[i];
    const size_t e = (size_t)(blockIdx.x - b.first[i]) * blockDim.x + threadIdx.x, nw = (size_t)w.n_out * w.k_in;
    if (e < nw) w.dW[e] = 0.f;
    else if (w.db && e - nw < (size_t)w.n_out) w.db[e - nw] = 0.f;
}

// NEFII_WGRAD_TR=0: keep the scalar-load kernel (A/B measurements)
static bool wgrad_tr_enabled() {
    static const bool v = [] {
        const char *e = getenv("NEFII_WGRAD_TR");
        return !(e && atoi(e) == 0);
    }();
    return v;
}

extern "C" int nefii_mlp_wgrad_f16(const float *dz, int dz_stride, const float *x, int x_stride, int64_t n, int n_out,
                                   int k_in, float scale, const float *gscale, float *dW, float *db, void *stream) {
    if (!dz || !x || !dW || !gscale || n_out <= 0 || k_in <= 0) return NEFII_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    int split = (int)((n + 255) / 256);       // <= 256 points per workgroup
    if (split < 1) split = 1;
    if (split > 64) split = 64;
    if (split > 1 || n <= 0) {
        const size_t nw = (size_t)n_out * k_in;
        const size_t nz = nw + (db ? (size_t)n_out : 0);
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, dW, nw, db, (size_t)n_out);
        HIP_CHECK_LAUNCH();
    }
    if (n <= 0) return 0;
    if (wgrad_tr_enabled() && n_out >= 64 && k_in >= 64 && n >= 1024 && (dz_stride & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(dz) & 15) == 0) {
        // blocked GEMM with transposed LDS reads; dW / db were zeroed above (n >= 1024 > 256 points: split > 1)
        int splits = (int)((n + WG2_POINTS - 1) / WG2_POINTS);
        splits = (splits + 7) / 8 * 8;
        const int tiles_n = (n_out + 127) / 128, tiles_k = (k_in + 127) / 128;
        const unsigned blocks = (unsigned)(tiles_n * tiles_k * splits);
        if ((x_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
            hipLaunchKernelGGL((mlp_wgrad16t_kernel<true, false, false>), dim3(blocks), dim3(256), 0, st, (const void *)dz, dz_stride,
                               (const void *)x, x_stride, n, n_out, k_in, scale, gscale, dW, db, splits, tiles_n, tiles_k);
        else
            hipLaunchKernelGGL((mlp_wgrad16t_kernel<false, false, false>), dim3(blocks), dim3(256), 0, st, (const void *)dz, dz_stride,
                               (const void *)x, x_stride, n, n_out, k_in, scale, gscale, dW, db, splits, tiles_n, tiles_k);
        HIP_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid((n_out + 63) / 64, (k_in + 255) / 256, split);
    hipLaunchKernelGGL((mlp_wgrad16_kernel<false, false>), grid, dim3(256), 0, st, (const void *)dz, dz_stride, (const void *)x,
                       x_stride, n, n_out, k_in, scale, gscale, dW, db, split > 1 ? 1 : 0);
    HIP_CHECK_LAUNCH();
    return 0;
}

// dz16 = [n][dz_stride] halves holding S dz (nefii_mlp_backward_f16h); x = fp32 rows (x_half = 0: the encoded network input
// of layer 0) or halves holding 16 h (x_half = 1: the forward's fp16 stash).  Strides in ELEMENTS of the operand's type.
extern "C" int nefii_mlp_wgrad_f16h(const void *dz16, int dz_stride, const void *x, int x_stride, int x_half, int64_t n,
                                    int n_out, int k_in, float scale, const float *gscale, float *dW, float *db, void *stream) {
    if (!dz16 || !x || !dW || !gscale || n_out <= 0 || k_in <= 0) return NEFII_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    int split = (int)((n + 255) / 256);       // <= 256 points per workgroup
    if (split < 1) split = 1;
    if (split > 64) split = 64;
    if (split > 1 || n <= 0) {
        const size_t nw = (size_t)n_out * k_in;
        const size_t nz = nw + (db ? (size_t)n_out : 0);
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, dW, nw, db, (size_t)n_out);
        HIP_CHECK_LAUNCH();
    }
    if (n <= 0) return 0;
    if (wgrad_tr_enabled() && n_out >= 64 && k_in >= 64 && n >= 1024 && (dz_stride & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(dz16) & 15) == 0) {
        int splits = (int)((n + WG2_POINTS - 1) / WG2_POINTS);
        splits = (splits + 7) / 8 * 8;
        const int tiles_n = (n_out + 127) / 128, tiles_k = (k_in + 127) / 128;
        const dim3 blocks((unsigned)(tiles_n * tiles_k * splits));
        const bool xvec = (x_stride & (x_half ? 7 : 3)) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#define NEFII_WG2H(XV, XH)                                                                                                    \
    hipLaunchKernelGGL((mlp_wgrad16t_kernel<XV, true, XH>), blocks, dim3(256), 0, st, dz16, dz_stride, x, x_stride, n, n_out, \
                       k_in, scale, gscale, dW, db, splits, tiles_n, tiles_k)
        if (x_half) {
            if (xvec) NEFII_WG2H(true, true);
            else NEFII_WG2H(false, true);
        } else {
            if (xvec) NEFII_WG2H(true, false);
            else NEFII_WG2H(false, false);
        }
#undef NEFII_WG2H
        HIP_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid((n_out + 63) / 64, (k_in + 255) / 256, split);
    if (x_half)
        hipLaunchKernelGGL((mlp_wgrad16_kernel<true, true>), grid, dim3(256), 0, st, dz16, dz_stride, x, x_stride, n, n_out, k_in,
                           scale, gscale, dW, db, split > 1 ? 1 : 0);
    else
        hipLaunchKernelGGL((mlp_wgrad16_kernel<true, false>), grid, dim3(256), 0, st, dz16, dz_stride, x, x_stride, n, n_out, k_in,
                           scale, gscale, dW, db, split > 1 ? 1 : 0);
    HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int nefii_mlp_wgrad_f16h_batch(const nefii_wgrad_item *h_items, int n_items, int64_t n, const float *gscale,
                                          void *stream) {
    if (!h_items || !gscale || n_items < 1 || n_items > NEFII_MAX_WGRAD_ITEMS) return NEFII_E_ARG;
    for (int i = 0; i < n_items; ++i) {
        const nefii_wgrad_item &w = h_items[i];
        if (!w.dz16 || !w.x || !w.dW || w.n_out <= 0 || w.k_in <= 0) return NEFII_E_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    int split = (int)((n + 255) / 256);       // the scalar-load kernel: <= 256 points per workgroup
    if (split < 1) split = 1;
    if (split > 64) split = 64;
    // one zero-fill for every accumulated output (the blocked kernel always accumulates, the scalar one when it splits the points)
    {
        WgradBatch z;
        z.n = 0;
        unsigned at = 0;
        for (int i = 0; i < n_items; ++i) {
            const nefii_wgrad_item &w = h_items[i];
            const bool blocked = wgrad_tr_enabled() && w.n_out >= 64 && w.k_in >= 64 && n >= 1024 && (w.dz_stride & 7) == 0 &&
                                 (reinterpret_cast<uintptr_t>(w.dz16) & 15) == 0;
            if (!(blocked || split > 1 || n <= 0)) continue;
            z.it[z.n] = w;
            z.first[z.n++] = at;
            at += (unsigned)(((size_t)w.n_out * w.k_in + (w.db ? w.n_out : 0) + 255) / 256);
        }
        z.first[z.n] = at;
        if (z.n > 0) {
            hipLaunchKernelGGL(zero_fill_batch_kernel, dim3(at), dim3(256), 0, st, z);
            HIP_CHECK_LAUNCH();
        }
    }
    if (n <= 0) return 0;
    // groups: (blocked GEMM | scalar-load kernel) x (x in halves | floats) x (x rows 16-byte loadable | not)
    for (int kind = 0; kind < 8; ++kind) {
        const bool want_blocked = kind & 4, want_half = kind & 2, want_vec = kind & 1;
        WgradBatch b;
        b.n = 0;
        unsigned at = 0;
        for (int i = 0; i < n_items; ++i) {
            const nefii_wgrad_item &w = h_items[i];
            const bool blocked = wgrad_tr_enabled() && w.n_out >= 64 && w.k_in >= 64 && n >= 1024 && (w.dz_stride & 7) == 0 &&
                                 (reinterpret_cast<uintptr_t>(w.dz16) & 15) == 0;
            const bool xvec = (w.x_stride & (w.x_half ? 7 : 3)) == 0 && (reinterpret_cast<uintptr_t>(w.x) & 15) == 0;
            if (blocked != want_blocked || (w.x_half != 0) != want_half) continue;
            if (blocked ? xvec != want_vec : want_vec) continue;      // (the scalar-load kernel has no vector form: one group)
            b.it[b.n] = w;
            b.first[b.n] = at;
            if (blocked) {
                int splits = (int)((n + WG2_POINTS - 1) / WG2_POINTS);
                splits = (splits + 7) / 8 * 8;
                b.aux0[b.n] = splits, b.aux1[b.n] = (w.n_out + 127) / 128, b.aux2[b.n] = (w.k_in + 127) / 128;
                at += (unsigned)(b.aux1[b.n] * b.aux2[b.n] * splits);
            } else {
                b.aux0[b.n] = (w.n_out + 63) / 64, b.aux1[b.n] = (w.k_in + 255) / 256, b.aux2[b.n] = split;
                at += (unsigned)(b.aux0[b.n] * b.aux1[b.n] * split);
            }
            ++b.n;
        }
        b.first[b.n] = at;
        if (b.n == 0) continue;
        if (want_blocked) {
            if (want_half && want_vec) hipLaunchKernelGGL((mlp_wgrad16t_batch_kernel<true, true>), dim3(at), dim3(256), 0, st, b, n, gscale);
            else if (want_half) hipLaunchKernelGGL((mlp_wgrad16t_batch_kernel<false, true>), dim3(at), dim3(256), 0, st, b, n, gscale);
            else if (want_vec) hipLaunchKernelGGL((mlp_wgrad16t_batch_kernel<true, false>), dim3(at), dim3(256), 0, st, b, n, gscale);
            else hipLaunchKernelGGL((mlp_wgrad16t_batch_kernel<false, false>), dim3(at), dim3(256), 0, st, b, n, gscale);
        } else {
            if (want_half) hipLaunchKernelGGL((mlp_wgrad16_batch_kernel<true>), dim3(at), dim3(256), 0, st, b, n, gscale);
            else hipLaunchKernelGGL((mlp_wgrad16_batch_kernel<false>), dim3(at), dim3(256), 0, st, b, n, gscale);
        }
        HIP_CHECK_LAUNCH();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// SDF value + d sdf / d x  (forward keeps the hidden activations in a workspace; the backward
// sweep runs in the same workgroup right after, so the workspace lines are still in L2)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void sdf_value_grad_kernel(nefii_mlp m, const float *__restrict__ x, int64_t n,
                                                                float *__restrict__ sdf_out, int out_stride,
                                                                float *__restrict__ feat_out, int feat_stride,
                                                                float *__restrict__ grad_out, float *__restrict__ ws,
                                                                int ws_stride) {
    __shared__ Lds lds;
    __shared__ float GE[TILE * ES];
    __shared__ float raw[TILE * 9];
    const int tid = threadIdx.x;
    const int Lm1 = m.n_layers - 1;
    int ke = 0;
    for (int l = 0; l < m.n_layers; ++l) ke = m.layer[l].k_e > ke ? m.layer[l].k_e : ke;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * TILE;
        load_tile_inputs(m, x, nullptr, nullptr, nullptr, base, n, raw, lds);
        for (int i = tid; i < TILE * ES; i += WG) GE[i] = 0.f;
        // ---- forward
        for (int l = 0; l <= Lm1; ++l) {
            const nefii_layer &L = m.layer[l];
            f32x16 acc[4];
            int ntw;
            layer_gemm(L, lds.X, lds.E, L.w_fwd, L.n_pad >> 5, acc, ntw);
            __syncthreads();
            NEFII_ACT_SWITCH(m.act, {
                NEFII_FOR_ACC(acc, ntw, {
                    const float z = val + L.bias[col];
                    const bool live = (base + row) < n;
                    if (l < Lm1) {
                        const float hval = act_fwd(z, ACT);
                        lds.X[row * XS + col] = hval;
                        if (live) ws[((size_t)l * n + base + row) * ws_stride + col] = hval;
                        if (l == Lm1 - 1 && feat_out && live && col < L.n_out)
                            feat_out[(size_t)(base + row) * feat_stride + col] = hval;
                    } else {
                        if (live && col < L.n_out) sdf_out[(size_t)(base + row) * out_stride + col] = z;
                        lds.X[row * XS + col] = (col == 0) ? 1.f : 0.f;   // seed d sdf / d z_{L-1}
                    }
                })
            })
            __syncthreads();
        }
        // ---- backward to the encoded input
        for (int l = Lm1; l >= 0; --l) {
            const nefii_layer &L = m.layer[l];
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            const int NTs = (L.k_x + L.k_e) >> 5;
            const float4 *wb = reinterpret_cast<const float4 *>(L.w_bwd);
            f32x16 acc[4];
            if (NTs > 16) {
                // the skip layer of a 512-wide net has 17-18 input tiles ([X | E] = 480 + 64): the tiles past the 16
                // that fit the accumulators are all encoding columns; do them first (they only touch GE, not lds.X)
                int ntb = (NTs - 16 - wave + 3) >> 2;
                if (ntb < 0) ntb = 0;
                zero_acc(acc);
                gemm_block(lds.X, XS, L.n_pad >> 3, wb + 16 * 64, NTs, wave, lane, ntb, acc);
                NEFII_FOR_ACC(acc, ntb, { GE[row * ES + (col + 512 - L.k_x)] += val; })
            }
            int ntw = ((NTs < 16 ? NTs : 16) - wave + 3) >> 2;
            if (ntw < 0) ntw = 0;
            zero_acc(acc);
            gemm_block(lds.X, XS, L.n_pad >> 3, wb, NTs, wave, lane, ntw, acc);
            __syncthreads();
            NEFII_ACT_SWITCH(m.act, {
                NEFII_FOR_ACC(acc, ntw, {
                    if (col < L.k_x) {
                        const bool live = (base + row) < n;
                        float v = 0.f;
                        if (live && l > 0) {
                            const float hprev = ws[((size_t)(l - 1) * n + base + row) * ws_stride + col];
                            v = val * act_bwd_from_out(hprev, ACT);
                        }
                        lds.X[row * XS + col] = v;
                    } else {
                        GE[row * ES + (col - L.k_x)] += val;
                    }
                })
            })
            __syncthreads();
        }
        // ---- chain through the positional encoding
        if (tid < TILE * 3) {
            const int p = tid / 3, c = tid - 3 * p;
            if (base + p < n) {
                const float *v = raw + p * 9;
                const int w0 = enc_width(m.enc_freqs[0]);
                float g = 0.f;
                for (int col = 0; col < w0; ++col) {
                    int comp;
                    const float d = enc_deriv(v, col, comp);
                    if (comp == c) g += GE[p * ES + col] * d;
                }
                grad_out[(size_t)(base + p) * 3 + c] = g;
            }
        }
        __syncthreads();
    }
}

// Split-precision variant (3 x fp16 MFMA per k-step, mlp_tile.h): the f32-input MFMA bounds the kernel above (52 % of
// its 157 TF at one tile per CU); on fp16 hi/lo operand pairs the same tile is bound by its weight stream instead.
__global__ __launch_bounds__(256, 1) void sdf_value_grad16_kernel(nefii_mlp m, const float *__restrict__ x, int64_t n,
                                                                  float *__restrict__ sdf_out, int out_stride,
                                                                  float *__restrict__ feat_out, int feat_stride,
                                                                  float *__restrict__ grad_out, float *__restrict__ ws,
                                                                  int ws_stride) {
    __shared__ Lds16 lds;
    __shared__ float GE[TILE * ES];
    __shared__ float raw[TILE * 9];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int Lm1 = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    int ke = 0;
    for (int l = 0; l < m.n_layers; ++l) ke = m.layer[l].k_e > ke ? m.layer[l].k_e : ke;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * TILE;
        for (int i = tid; i < TILE * 9; i += WG) {
            const int p = i / 9, c = i - 9 * p;
            int64_t idx = base + p;
            if (idx >= n) idx = n - 1;
            raw[i] = c < 3 ? x[idx * 3 + c] : 0.f;
        }
        for (int i = tid; i < TILE * ES; i += WG) GE[i] = 0.f;
        __syncthreads();
        encode_tile16(m, raw, lds, ke);
        __syncthreads();
        // ---- forward
        for (int l = 0; l <= Lm1; ++l) {
            const nefii_layer &L = m.layer[l];
            f32x16 acc[4];
            int ntw;
            layer_gemm16(L, lds, L.n_pad >> 5, acc, ntw);
            __syncthreads();
            NEFII_ACT_SWITCH(m.act, {
                NEFII_FOR_ACC(acc, ntw, {
                    const float z = val * inv_scale + L.bias[col];
                    const bool live = (base + row) < n;
                    if (l < Lm1) {
                        const float hval = ACT == NEFII_ACT_SOFTPLUS100 ? softplus100_s16(z * A16_SCALE) * (1.f / A16_SCALE) : act_fwd(z, ACT);
                        split16a(hval, lds.Xh[row * XS16 + col], lds.Xl[row * XS16 + col]);
                        if (live) ws[((size_t)l * n + base + row) * ws_stride + col] = hval;
                        if (l == Lm1 - 1 && feat_out && live && col < L.n_out)
                            feat_out[(size_t)(base + row) * feat_stride + col] = hval;
                    } else {
                        if (live && col < L.n_out) sdf_out[(size_t)(base + row) * out_stride + col] = z;
                        split16a(col == 0 ? 1.f : 0.f, lds.Xh[row * XS16 + col], lds.Xl[row * XS16 + col]);   // seed
                    }
                })
            })
            __syncthreads();
        }
        // ---- backward to the encoded input
        for (int l = Lm1; l >= 0; --l) {
            const nefii_layer &L = m.layer[l];
            const int NTs = (L.k_x + L.k_e) >> 5;
            const half8 *wb = reinterpret_cast<const half8 *>(L.w_bwd_f16x3);
            f32x16 acc[4];
            if (NTs > 16) {     // input tiles past the 16 the accumulators hold are encoding columns: GE only
                int ntb = (NTs - 16 - wave + 3) >> 2;
                if (ntb < 0) ntb = 0;
                zero_acc(acc);
                gemm_block16(lds.Xh, lds.Xl, XS16, L.n_pad >> 4, wb + 16 * 2 * 64, NTs, wave, lane, ntb, acc);
                NEFII_FOR_ACC(acc, ntb, { GE[row * ES + (col + 512 - L.k_x)] += val * inv_scale; })
            }
            int ntw = ((NTs < 16 ? NTs : 16) - wave + 3) >> 2;
            if (ntw < 0) ntw = 0;
            zero_acc(acc);
            gemm_block16(lds.Xh, lds.Xl, XS16, L.n_pad >> 4, wb, NTs, wave, lane, ntw, acc);
            __syncthreads();
            NEFII_ACT_SWITCH(m.act, {
                NEFII_FOR_ACC(acc, ntw, {
                    const float g = val * inv_scale;
                    if (col < L.k_x) {
                        const bool live = (base + row) < n;
                        float v = 0.f;
                        if (live && l > 0) {
                            const float hprev = ws[((size_t)(l - 1) * n + base + row) * ws_stride + col];
                            v = g * (ACT == NEFII_ACT_SOFTPLUS100 ? softplus100_bwd_fast(hprev) : act_bwd_from_out(hprev, ACT));
                        }
                        split16a(v, lds.Xh[row * XS16 + col], lds.Xl[row * XS16 + col]);
                    } else {
                        GE[row * ES + (col - L.k_x)] += g;
                    }
                })
            })
            __syncthreads();
        }
        // ---- chain through the positional encoding
        if (tid < TILE * 3) {
            const int p = tid / 3, c = tid - 3 * p;
            if (base + p < n) {
                const float *v = raw + p * 9;
                const int w0 = enc_width(m.enc_freqs[0]);
                float g = 0.f;
                for (int col = 0; col < w0; ++col) {
                    int comp;
                    const float d = enc_deriv(v, col, comp);
                    if (comp == c) g += GE[p * ES + col] * d;
                }
                grad_out[(size_t)(base + p) * 3 + c] = g;
            }
        }
        __syncthreads();
    }
}

static int sdf_ws_stride(const nefii_mlp *m) {
    int s = 32;
    for (int l = 0; l < m->n_layers - 1; ++l) s = m->layer[l].n_pad > s ? m->layer[l].n_pad : s;
    return s;
}

extern "C" size_t nefii_sdf_value_grad_workspace_bytes(const nefii_mlp *h_mlp, int64_t n) {
    if (!h_mlp || n <= 0) return 0;
    if (const size_t b = value_grad_stream_ws_bytes(h_mlp, n)) return b;     // the streamed kernel's per-workgroup slots
    return (size_t)(h_mlp->n_layers - 1) * (size_t)n * sdf_ws_stride(h_mlp) * sizeof(float);
}

extern "C" int nefii_sdf_value_grad(const nefii_mlp *h_mlp, const float *x, int64_t n, float *sdf_out, int out_stride,
                                    float *feat_out, int feat_stride, float *grad_out, float *ws, void *stream) {
    int rc = check_mlp(h_mlp);
    if (rc) return rc;
    if (n <= 0) return 0;
    if (!x || !sdf_out || !grad_out || !ws) return NEFII_E_ARG;
    if (h_mlp->enc_freqs[0] < 0 || h_mlp->enc_freqs[1] >= 0 || h_mlp->enc_freqs[2] >= 0 || h_mlp->feat_width != 0)
        return NEFII_E_UNSUPPORTED;
    bool split = true;
    for (int l = 0; l < h_mlp->n_layers; ++l)
        split = split && h_mlp->layer[l].w_f16x3 && h_mlp->layer[l].w_bwd_f16x3;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    if (split && value_grad_stream_ws_bytes(h_mlp, n))      // 512-wide nets with a fragment stream: the pipelined kernel
        return value_grad_stream_launch(h_mlp, x, n, sdf_out, out_stride, feat_out, feat_stride, grad_out, ws,
                                        (hipStream_t)stream);
    if (split) {        // both fp16 hi/lo fragment sets present: the split-precision kernel
        hipLaunchKernelGGL(sdf_value_grad16_kernel, dim3(grid_for(n_tiles, 1)), dim3(WG), 0, (hipStream_t)stream,
                           *h_mlp, x, n, sdf_out, out_stride, feat_out, feat_stride, grad_out, ws,
                           sdf_ws_stride(h_mlp));
        HIP_CHECK_LAUNCH();
        return 0;
    }
    for (int l = 0; l < h_mlp->n_layers; ++l)
        if (!h_mlp->layer[l].w_bwd) return NEFII_E_ARG;
    hipLaunchKernelGGL(sdf_value_grad_kernel, dim3(grid_for(n_tiles, 1)), dim3(WG), 0, (hipStream_t)stream, *h_mlp, x,
                       n, sdf_out, out_stride, feat_out, feat_stride, grad_out, ws, sdf_ws_stride(h_mlp));
    HIP_CHECK_LAUNCH();
    return 0;
}
