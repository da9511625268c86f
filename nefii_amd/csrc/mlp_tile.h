// mlp_tile.h - workgroup-level fused-MLP building blocks for gfx950 (MI355X, CDNA4).
//
// One workgroup = 256 threads = 4 wave64 = one tile of 32 points.  A layer is the GEMM
//     Y[32 x n_pad] = [ X[32 x k_x] | E[32 x k_e] ] * W^T
// done on the f32-input matrix core (v_mfma_f32_32x32x2_f32: exact fp32 fma chains - the SDF
// decides ray/surface intersections against a 5e-5 threshold, so the tracer cannot afford
// 16-bit operands; BASELINE.md precision table).  Data placement:
//   * activations live in LDS ("X": hidden/feature block, row stride 516 floats; "E": encoded raw
//     inputs, row stride 100 floats).  Strides are 4*odd so that the ds_read_b128 A-fragment
//     reads (row = lane&31, 4 consecutive k) are bank-conflict free (MI355X_MICROARCH.md, LDS);
//   * weights are NOT staged in LDS: each wave owns its own 32-column output tiles, so a weight
//     element is used by exactly one wave - it is streamed L2 -> VGPR as one coalesced
//     global_load_dwordx4 per lane (1 KiB per wave instruction) from a host-packed fragment order,
//     double-buffered in registers against the 64-cycle MFMAs;
//   * accumulators: up to 4 tiles x 16 VGPR per wave.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include "../../include/nefii_amd.h"

namespace nefii {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The 8-wave streaming kernels (the tracer's evaluators, which TrainStep runs on side streams beside the current batch's
// tail, and the tail's own tile kernels) claim their SIMDs' whole vector register file, so that no wave of another
// kernel is ever placed on a SIMD that hosts their MFMA waves.  (The 4-wave kernels - blocked weight gradients, the 32-row
// f32 / fp16 MLP kernels - do not: two of their workgroups share a CU by design and a lone one leaves registers free;
// they only ever run on the caller's stream, beside nothing but the tracer's kernels, which are compiled without
// packed-fp32 instructions.)  Found in round 3 (tools/concurrency_probe.py, DESIGN.md
// "Packed fp32 beside MFMA waves"): on gfx950 a packed-fp32 VALU instruction with op_sel set on src1 (v_pk_mul_f32 /
// v_pk_add_f32 / v_pk_fma_f32 ... op_sel:[0,1] - hipcc forms them from adjacent scalar float operations, e.g. a cross product)
// reads ZERO for src1 in the low result lane of lanes 48-63, silently, while its wave shares a SIMD with waves of the
// single-pass evaluator - alone, or beside kernels that fill the register file, it is exact.  A 209-register
// evaluator wave left room for one 80-register wave of the MC sampler: config 3 got non-finite pdfs in a third of its steps.
// The claim costs the kernel nothing (LDS already limits it to these waves); other kernels get CUs that host none of its
// workgroups.  The library itself is compiled without packed-fp32 instructions (nefii_amd/build.py), so its own VALU code
// is never the victim; the claim protects the kernels it does not compile (torch's).
//   NEFII_CLAIM_SIMD_2: two waves of this kernel per SIMD (8-wave workgroups, one per CU) - 256 registers each;
//   NEFII_CLAIM_SIMD_1: one wave per SIMD (4-wave workgroups) - all 512 (256 architectural + 256 accumulation).
#ifdef NEFII_NO_CLAIM       /* A/B builds only (tools/concurrency_probe.py: which instruction forms are victims) */
#define NEFII_CLAIM_SIMD_2()
#define NEFII_CLAIM_SIMD_1()
#else
#define NEFII_CLAIM_SIMD_2() asm volatile("" ::: "v255")
#define NEFII_CLAIM_SIMD_1() asm volatile("" ::: "v255", "a255")
#endif

constexpr int TILE = NEFII_TILE_ROWS;   // 32 rows
constexpr int XS = 516;                 // X row stride (floats): 4*129
constexpr int ES = 100;                 // E row stride (floats): 4*25
constexpr int WG = 256;

struct Lds {
    float X[TILE * XS];
    float E[TILE * ES];
};

// Softplus(beta=100, threshold=20) = log1p(exp(100 v))/100 (torch.nn.Softplus, implicit_differentiable_renderer.py:83)
// on the hardware transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each) with the two places
// where a plain exp2/log2 formulation loses bits compensated:
//   * exp(z) = 2^(z*log2e): the rounding error of the product z*log2e (up to |z| * 2^-24) is carried as t_lo
//     and folded back in first order, so exp keeps ~1 ulp for the whole range of z;
//   * log1p(e) = log(u) + (e - (u-1))/u with u = fl(1+e): exact to first order in the rounding of 1+e, and
//     reduces to e for e < 2^-24.
// ~16 VALU instructions instead of ~150 for the libm expf/log1pf + IEEE division pair (the epilogue of a
// 512-wide layer evaluates 64 of these per lane).
__device__ __forceinline__ float softplus100(float v) {
    const float z = v * 100.f;
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f, LN2 = 0.693147182464599609375f;
    const float t_hi = z * L2E_HI;
    const float t_lo = __builtin_fmaf(z, L2E_HI, -t_hi) + z * L2E_LO;
    const float e_hw = __builtin_amdgcn_exp2f(t_hi);
    const float e = __builtin_fmaf(e_hw, t_lo * LN2, e_hw);
    const float u = 1.f + e;
    const float c = (e - (u - 1.f)) * __builtin_amdgcn_rcpf(u);
    const float r = __builtin_fmaf(__builtin_amdgcn_logf(u), LN2, c);     // v_log_f32 is log2
    return z > 20.f ? v : r * 0.01f;
}

// Runs BODY with ACT a compile-time copy of the (wave-uniform) activation id.  Epilogues apply the activation to 64
// values per lane; with the id tested per value every value took its own scalar branch and the compiler could not
// interleave the values' dependent exp/log chains.
#define NEFII_ACT_SWITCH(act, BODY)                                  \
    if ((act) == NEFII_ACT_SOFTPLUS100) {                            \
        constexpr int ACT = NEFII_ACT_SOFTPLUS100;                   \
        BODY                                                         \
    } else if ((act) == NEFII_ACT_ELU) {                             \
        constexpr int ACT = NEFII_ACT_ELU;                           \
        BODY                                                         \
    } else {                                                         \
        constexpr int ACT = NEFII_ACT_RELU;                          \
        BODY                                                         \
    }

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == NEFII_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == NEFII_ACT_ELU) return v > 0.f ? v : expm1f(v);
    return softplus100(v);
}

// derivative of the activation expressed through its OUTPUT h (what the stash keeps)
__device__ __forceinline__ float act_bwd_from_out(float h, int act) {
    if (act == NEFII_ACT_RELU) return h > 0.f ? 1.f : 0.f;
    if (act == NEFII_ACT_ELU) return h > 0.f ? 1.f : h + 1.f;
    // softplus: h = log1p(e^{100 z})/100  =>  sigmoid(100 z) = 1 - e^{-100 h}
    return -expm1f(-100.f * h);
}

// d Softplus(beta 100) / dz from the activation's OUTPUT h: sigmoid(100 z) = 1 - e^{-100 h}, on v_exp_f32 (the libm
// expm1f of act_bwd_from_out costs ~40 instructions; absolute error here ~1e-7, the result multiplies a gradient)
__device__ __forceinline__ float softplus100_bwd_fast(float h) {
    return 1.f - __builtin_amdgcn_exp2f(h * (-100.f * 1.44269504088896340736f));
}

__device__ __forceinline__ float head_fwd(float v, int head) {
    switch (head) {
        case NEFII_HEAD_TANH01: return (tanhf(v) + 1.f) / 2.f;
        case NEFII_HEAD_POW2: return v * v;
        case NEFII_HEAD_SIGMOID: return 1.f / (1.f + expf(-v));
        case NEFII_HEAD_RELU: return v > 0.f ? v : 0.f;
        case NEFII_HEAD_ABS: return fabsf(v);
        case NEFII_HEAD_RELU_INIT: return (v > 0.f ? v : 0.f) + 0.5f;
        default: return v;
    }
}

// d head / d pre-activation, from the head OUTPUT y (and, where y is not enough, the sign convention below)
__device__ __forceinline__ float head_bwd_from_out(float y, float pre, int head) {
    switch (head) {
        case NEFII_HEAD_TANH01: { float t = 2.f * y - 1.f; return 0.5f * (1.f - t * t); }
        case NEFII_HEAD_POW2: return 2.f * pre;
        case NEFII_HEAD_SIGMOID: return y * (1.f - y);
        case NEFII_HEAD_RELU: return pre > 0.f ? 1.f : 0.f;
        case NEFII_HEAD_ABS: return pre > 0.f ? 1.f : (pre < 0.f ? -1.f : 0.f);
        case NEFII_HEAD_RELU_INIT: return pre > 0.f ? 1.f : 0.f;
        default: return 1.f;
    }
}

__device__ __forceinline__ int enc_width(int L) { return L < 0 ? 0 : 3 + 6 * L; }

// value of column c of the positional encoding of v (embedder.py:21-31 order: x, sin f0, cos f0, sin f1, ...)
__device__ __forceinline__ float enc_value(const float v[3], int c) {
    if (c < 3) return v[c];
    int q = c - 3;
    int k = q / 6, rem = q - 6 * k;
    int fn = rem / 3, comp = rem - 3 * fn;
    float a = v[comp] * (float)(1 << k);
    return fn ? cosf(a) : sinf(a);
}

// derivative of column c wrt its own component (and which component it is)
__device__ __forceinline__ float enc_deriv(const float v[3], int c, int &comp) {
    if (c < 3) { comp = c; return 1.f; }
    int q = c - 3;
    int k = q / 6, rem = q - 6 * k;
    int fn = rem / 3;
    comp = rem - 3 * fn;
    float f = (float)(1 << k);
    float a = v[comp] * f;
    return fn ? -f * sinf(a) : f * cosf(a);
}

// Fill E[32][k_e] from up to three raw [n,3] inputs already staged in LDS `raw` ([32][9]).
__device__ __forceinline__ void encode_tile(const nefii_mlp &m, const float *raw, float *E, int k_e) {
    const int tid = threadIdx.x;
    const int p = tid & 31, part = tid >> 5;
    const int w0 = enc_width(m.enc_freqs[0]), w1 = enc_width(m.enc_freqs[1]), w2 = enc_width(m.enc_freqs[2]);
    for (int c = part; c < k_e; c += 8) {
        float val = 0.f;
        if (c < w0) {
            val = enc_value(raw + p * 9, c);
        } else if (c < w0 + w1) {
            val = enc_value(raw + p * 9 + 3, c - w0);
        } else if (c < w0 + w1 + w2) {
            val = enc_value(raw + p * 9 + 6, c - w0 - w1);
        }
        E[p * ES + c] = val;
    }
}

// ---- GEMM core -------------------------------------------------------------------------------
// acc[j] += A[32 x 8*kgroups] * Wfrag for this wave's tiles t = wave + 4*j (j < ntw).
// A fragment: lane (r = lane&31, h = lane>>5) reads A[r][8g + 4h .. +3] with one ds_read_b128;
// MFMA step s multiplies k = 8g+4h+s: over h (the MFMA's own K=2) and s = 0..3 all 8 k of the group.
// wp points at group 0 of this layer block: float4 index ((g*NT + t)*64 + lane).
#ifndef NEFII_FD
#define NEFII_FD 6
#endif
template <int NTW>
__device__ __forceinline__ void gemm_block_t(const float *arow, int kgroups, const float4 *__restrict__ wp, int NT,
                                             int wave, int lane, f32x16 (&acc)[4]) {
    // NB statically indexed register stages of weight fragments (NB-1 k-groups in flight towards L2: the kernels are
    // latency/bandwidth bound on that stream), two stages of the LDS A fragment; k-loop unrolled NB times so that no
    // stage is ever copied.
    constexpr int NB = NEFII_FD;
    static_assert(NB >= 2 && NB % 2 == 0, "NB must be even");
    float4 b[NB][NTW];
    float4 a[2];
    auto load_b = [&](int u, int g) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) b[u][j] = wp[((size_t)g * NT + wave + 4 * j) * 64 + lane];
    };
#pragma unroll
    for (int u = 0; u < NB - 1; ++u)
        if (u < kgroups) load_b(u, u);
    a[0] = *reinterpret_cast<const float4 *>(arow);
    for (int g0 = 0; g0 < kgroups; g0 += NB) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int g = g0 + u;
            if (g < kgroups) {
                if (g + NB - 1 < kgroups) load_b((u + NB - 1) % NB, g + NB - 1);
                if (g + 1 < kgroups) a[(u + 1) % 2] = *reinterpret_cast<const float4 *>(arow + 8 * (g + 1));
                const float4 av = a[u % 2];
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[u][j].x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[u][j].y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[u][j].z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[u][j].w, acc[j], 0, 0, 0);
                }
            }
        }
    }
}

__device__ __forceinline__ void gemm_block(const float *A, int a_stride, int kgroups, const float4 *__restrict__ wp,
                                           int NT, int wave, int lane, int ntw, f32x16 (&acc)[4]) {
    if (kgroups <= 0 || ntw <= 0) return;
    const int r = lane & 31, h = lane >> 5;
    const float *arow = A + r * a_stride + 4 * h;
    switch (ntw) {          // ntw is wave-uniform; each case has fully static register indexing
        case 1: gemm_block_t<1>(arow, kgroups, wp, NT, wave, lane, acc); break;
        case 2: gemm_block_t<2>(arow, kgroups, wp, NT, wave, lane, acc); break;
        case 3: gemm_block_t<3>(arow, kgroups, wp, NT, wave, lane, acc); break;
        default: gemm_block_t<4>(arow, kgroups, wp, NT, wave, lane, acc); break;   // callers pass ntw <= 4
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
}

// One layer's matrix product for the whole workgroup: acc = [X | E] * W.  Caller does the epilogue.
__device__ __forceinline__ void layer_gemm(const nefii_layer &L, const float *X, const float *E, const float *w,
                                           int n_tiles, f32x16 (&acc)[4], int &ntw) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    ntw = (n_tiles - wave + 3) >> 2;
    if (ntw < 0) ntw = 0;
    zero_acc(acc);
    const float4 *wp = reinterpret_cast<const float4 *>(w);
    gemm_block(X, XS, L.k_x >> 3, wp, n_tiles, wave, lane, ntw, acc);
    gemm_block(E, ES, L.k_e >> 3, wp + (size_t)(L.k_x >> 3) * n_tiles * 64, n_tiles, wave, lane, ntw, acc);
}

// ================================================================================================
// Split-precision variant: operands as fp16 (hi, lo) pairs, three v_mfma_f32_32x32x16_f16 per 16-deep k-step
//     x*w ~= xh*wh + xh*wl + xl*wh      (fp32 accumulate; every partial product is exact in fp32)
// ~5.3x the f32-input MFMA rate with ~2^-22 relative operand error, i.e. fp32-class accuracy (a single fp16 or
// bf16 pass fails the 1e-3 parity bar: BASELINE.md).  Activations live in LDS as two half arrays (row stride
// 520 halves = 16 B * 65: ds_read_b128 of the A fragment stays conflict-free), weights stream from L2 in
// 32x32x16 fragment order, hi block then lo block, pre-scaled by 64 (exact) so that lo halves stay normal.
// ================================================================================================
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int XS16 = 520;               // halves per X row
constexpr int ES16 = 104;               // halves per E row (16 B * 13)
constexpr float W16_SCALE = 64.f;
// activations are split as (16 x) so that the lo half of small values (softplus(~0)/100 ~ 7e-3) is a normal fp16
// number instead of a subnormal with 6e-8 resolution; exact power-of-two scaling, undone with the weight scale
constexpr float A16_SCALE = 16.f;

struct Lds16 {
    _Float16 Xh[TILE * XS16], Xl[TILE * XS16];
    _Float16 Eh[TILE * ES16], El[TILE * ES16];
};

__device__ __forceinline__ void split16(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
__device__ __forceinline__ void split16a(float v, _Float16 &hi, _Float16 &lo) { split16(v * A16_SCALE, hi, lo); }

__device__ __forceinline__ void encode_tile16(const nefii_mlp &m, const float *raw, Lds16 &lds, int k_e) {
    const int tid = threadIdx.x;
    const int p = tid & 31, part = tid >> 5;
    const int w0 = enc_width(m.enc_freqs[0]), w1 = enc_width(m.enc_freqs[1]), w2 = enc_width(m.enc_freqs[2]);
    for (int c = part; c < k_e; c += 8) {
        float val = 0.f;
        if (c < w0) {
            val = enc_value(raw + p * 9, c);
        } else if (c < w0 + w1) {
            val = enc_value(raw + p * 9 + 3, c - w0);
        } else if (c < w0 + w1 + w2) {
            val = enc_value(raw + p * 9 + 6, c - w0 - w1);
        }
        split16a(val, lds.Eh[p * ES16 + c], lds.El[p * ES16 + c]);
    }
}

// acc[j] += A[32 x 16*ksteps] * W for this wave's tiles t = wave + 4 j (j < NTW).
// wp: half8 index ((s*NT + t)*2 + part)*64 + lane.  NB statically named register stages of weight fragments (NB-1
// k-steps in flight: these kernels run one 32-row tile per CU and are bound by how fast one workgroup pulls its
// 1 MiB per layer from L2), two stages of the activation fragments; the k-loop is unrolled NB times so that no stage is
// ever copied.  Kernels using it run one workgroup of 4 waves per CU (512 registers per wave).
#ifndef NEFII_HD
#define NEFII_HD 4       /* measured on sdf_value_grad16: 4 / 6 / 8 stages -> 0.551 / 0.550 / 0.663 ms (8 spills) */
#endif
// SP: single pass - hi fragments only, one MFMA per k-step and tile (the radiance / material MLPs' fp16 kernels; `al` unused)
template <int NTW, bool SP = false>
__device__ __forceinline__ void gemm_block16_t(const _Float16 *ah, const _Float16 *al, int ksteps,
                                               const half8 *__restrict__ wp, int NT, int wave, int lane,
                                               f32x16 (&acc)[4]) {
    constexpr int NB = NEFII_HD;
    static_assert(NB >= 2 && NB % 2 == 0, "NB must be even");
    half8 bh[NB][NTW], bl[SP ? 1 : NB][SP ? 1 : NTW];
    half8 a_hi[2], a_lo[SP ? 1 : 2];
    auto load_b = [&](int u, int s) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const size_t t = (size_t)s * NT + wave + 4 * j;
            bh[u][j] = wp[(t * 2) * 64 + lane];
            if constexpr (!SP) bl[u][j] = wp[(t * 2 + 1) * 64 + lane];
        }
    };
#pragma unroll
    for (int u = 0; u < NB - 1; ++u)
        if (u < ksteps) load_b(u, u);
    a_hi[0] = *reinterpret_cast<const half8 *>(ah);
    if constexpr (!SP) a_lo[0] = *reinterpret_cast<const half8 *>(al);
    for (int s0 = 0; s0 < ksteps; s0 += NB) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int s = s0 + u;
            if (s < ksteps) {
                if (s + NB - 1 < ksteps) load_b((u + NB - 1) % NB, s + NB - 1);
                if (s + 1 < ksteps) {
                    a_hi[(u + 1) % 2] = *reinterpret_cast<const half8 *>(ah + 16 * (s + 1));
                    if constexpr (!SP) a_lo[(u + 1) % 2] = *reinterpret_cast<const half8 *>(al + 16 * (s + 1));
                }
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[u % 2], bh[u][j], acc[j], 0, 0, 0);
                    if constexpr (!SP) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[u % 2], bl[u][j], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[u % 2], bh[u][j], acc[j], 0, 0, 0);
                    }
                }
            }
        }
    }
}

template <bool SP = false>
__device__ __forceinline__ void gemm_block16(const _Float16 *Ah, const _Float16 *Al, int a_stride, int ksteps,
                                             const half8 *__restrict__ wp, int NT, int wave, int lane, int ntw,
                                             f32x16 (&acc)[4]) {
    if (ksteps <= 0 || ntw <= 0) return;
    const int r = lane & 31, h = lane >> 5;
    const _Float16 *ah = Ah + r * a_stride + 8 * h;
    const _Float16 *al = Al + r * a_stride + 8 * h;
    switch (ntw) {          // ntw is wave-uniform; each case has fully static register indexing
        case 1: gemm_block16_t<1, SP>(ah, al, ksteps, wp, NT, wave, lane, acc); break;
        case 2: gemm_block16_t<2, SP>(ah, al, ksteps, wp, NT, wave, lane, acc); break;
        case 3: gemm_block16_t<3, SP>(ah, al, ksteps, wp, NT, wave, lane, acc); break;
        default: gemm_block16_t<4, SP>(ah, al, ksteps, wp, NT, wave, lane, acc); break;   // callers pass ntw <= 4
    }
}

// single-pass tile of the radiance / material kernels: hi halves of activations and encoded inputs only
struct Lds16h {
    _Float16 Xh[TILE * XS16];
    _Float16 Eh[TILE * ES16];
};

__device__ __forceinline__ void layer_gemm16h(const nefii_layer &L, const Lds16h &lds, int n_tiles, f32x16 (&acc)[4],
                                              int &ntw) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    ntw = (n_tiles - wave + 3) >> 2;
    if (ntw < 0) ntw = 0;
    zero_acc(acc);
    const half8 *wp = reinterpret_cast<const half8 *>(L.w_f16x3);
    gemm_block16<true>(lds.Xh, lds.Xh, XS16, L.k_x >> 4, wp, n_tiles, wave, lane, ntw, acc);
    gemm_block16<true>(lds.Eh, lds.Eh, ES16, L.k_e >> 4, wp + (size_t)(L.k_x >> 4) * n_tiles * 2 * 64, n_tiles, wave, lane,
                       ntw, acc);
}

__device__ __forceinline__ void layer_gemm16(const nefii_layer &L, const Lds16 &lds, int n_tiles, f32x16 (&acc)[4],
                                             int &ntw) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ntw = (n_tiles - wave + 3) >> 2;
    if (ntw < 0) ntw = 0;
    zero_acc(acc);
    const half8 *wp = reinterpret_cast<const half8 *>(L.w_f16x3);
    gemm_block16(lds.Xh, lds.Xl, XS16, L.k_x >> 4, wp, n_tiles, wave, lane, ntw, acc);
    gemm_block16(lds.Eh, lds.El, ES16, L.k_e >> 4, wp + (size_t)(L.k_x >> 4) * n_tiles * 2 * 64, n_tiles, wave, lane,
                 ntw, acc);
}

// ---- wide variant: 64 rows per workgroup, 8 waves; a wave owns a 64-column pair of tiles for BOTH 32-row
// halves, so every weight fragment it pulls from L2 feeds two MFMA row tiles (half the L2 bytes per query:
// the 32-row kernel is bound by the ~41 GB/s/CU it can stream from L2, not by the matrix cores).
constexpr int TILE_W = 64;
constexpr int WG_W = 512;
struct Lds16w {
    _Float16 Xh[TILE_W * XS16], Xl[TILE_W * XS16];
    _Float16 Eh[TILE_W * ES16], El[TILE_W * ES16];
};

__device__ __forceinline__ void encode_tile16w(const nefii_mlp &m, const float *raw, Lds16w &lds, int k_e) {
    const int tid = threadIdx.x;
    const int p = tid & 63, part = tid >> 6;
    const int w0 = enc_width(m.enc_freqs[0]), w1 = enc_width(m.enc_freqs[1]), w2 = enc_width(m.enc_freqs[2]);
    for (int c = part; c < k_e; c += 8) {
        float val = 0.f;
        if (c < w0) {
            val = enc_value(raw + p * 9, c);
        } else if (c < w0 + w1) {
            val = enc_value(raw + p * 9 + 3, c - w0);
        } else if (c < w0 + w1 + w2) {
            val = enc_value(raw + p * 9 + 6, c - w0 - w1);
        }
        split16a(val, lds.Eh[p * ES16 + c], lds.El[p * ES16 + c]);
    }
}

// acc[rt*2 + ct] += A[rows 32rt..][16*ksteps] * W[tiles 2*wave + ct]   (nct = column tiles of this wave: 0..2)
// Three statically named register stages for the weight fragments (two k-steps in flight) and for the A fragments
// (one k-step ahead); the k-loop is unrolled by three so that no stage is ever copied (a rotating buffer costs
// 48 v_mov per k-step, a third of the MFMA issue time).
// Register pipeline of the wide kernel: NB statically named stages of weight fragments (NB-1 k-steps in flight:
// the kernel is bound by how many bytes per CU it keeps in flight towards L2 - Little's law; measured on config 2:
// NB 2 / 4 / 6 -> 9.25 / 8.94 / 8.39 ms of eval time per step) and two stages of activation fragments (one k-step
// ahead, LDS latency).  The k-loop is unrolled NB times so that every stage index is a compile-time constant - a
// rotating buffer would cost ~48 v_mov per k-step.
#ifndef NEFII_WD
#define NEFII_WD 6
#endif
struct BStage16w {
    half8 bh[2], bl[2];      // weight fragments of the wave's two column tiles
};
struct AStage16w {
    half8 ah[2], al[2];      // activation fragments of the two row tiles
};

template <bool TWO_COLS>
__device__ __forceinline__ void load_b16w(BStage16w &st, const half8 *__restrict__ wp, int NT, int tile0, int lane,
                                          int s) {
    const size_t t = (size_t)s * NT + tile0;
    st.bh[0] = wp[(t * 2) * 64 + lane];
    st.bl[0] = wp[(t * 2 + 1) * 64 + lane];
    if (TWO_COLS) {
        st.bh[1] = wp[((t + 1) * 2) * 64 + lane];
        st.bl[1] = wp[((t + 1) * 2 + 1) * 64 + lane];
    }
}

__device__ __forceinline__ void load_a16w(AStage16w &st, const _Float16 *ah, const _Float16 *al, int rt_off, int s) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        st.ah[rt] = *reinterpret_cast<const half8 *>(ah + rt * rt_off + 16 * s);
        st.al[rt] = *reinterpret_cast<const half8 *>(al + rt * rt_off + 16 * s);
    }
}

template <bool TWO_COLS>
__device__ __forceinline__ void mfma16w(const AStage16w &sa, const BStage16w &sb, f32x16 (&acc)[4]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < (TWO_COLS ? 2 : 1); ++c) {
            // weights are the MFMA's A operand, activations its B operand: the accumulator holds the TRANSPOSED
            // product, lane = query (lane&31), registers = 4 consecutive features x 4 groups - so the epilogue packs
            // four halves of one LDS row per store and loads biases as float4
            f32x16 &a = acc[rt * 2 + c];
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(sb.bh[c], sa.ah[rt], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(sb.bl[c], sa.ah[rt], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(sb.bh[c], sa.al[rt], a, 0, 0, 0);
        }
}

template <bool TWO_COLS>
__device__ __forceinline__ void gemm_block16w_t(const _Float16 *Ah, const _Float16 *Al, int a_stride, int ksteps,
                                                const half8 *__restrict__ wp, int NT, int wave, int lane,
                                                f32x16 (&acc)[4]) {
    constexpr int NB = NEFII_WD;
    static_assert(NB >= 2 && NB % 2 == 0, "NB must be even (activation stages alternate)");
    const int r = lane & 31, h = lane >> 5;
    const _Float16 *ah = Ah + r * a_stride + 8 * h;
    const _Float16 *al = Al + r * a_stride + 8 * h;
    const int rt_off = 32 * a_stride;
    BStage16w b[NB];
    AStage16w a[2];
#pragma unroll
    for (int u = 0; u < NB - 1; ++u)
        if (u < ksteps) load_b16w<TWO_COLS>(b[u], wp, NT, wave, lane, u);
    load_a16w(a[0], ah, al, rt_off, 0);
    for (int s0 = 0; s0 < ksteps; s0 += NB) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int s = s0 + u;
            if (s < ksteps) {
                if (s + NB - 1 < ksteps) load_b16w<TWO_COLS>(b[(u + NB - 1) % NB], wp, NT, wave, lane, s + NB - 1);
                if (s + 1 < ksteps) load_a16w(a[(u + 1) % 2], ah, al, rt_off, s + 1);
                mfma16w<TWO_COLS>(a[u % 2], b[u], acc);
            }
        }
    }
}

__device__ __forceinline__ void gemm_block16w(const _Float16 *Ah, const _Float16 *Al, int a_stride, int ksteps,
                                              const half8 *__restrict__ wp, int NT, int tile0, int lane, int nct,
                                              f32x16 (&acc)[4]) {
    if (ksteps <= 0 || nct <= 0) return;
    if (nct == 2)
        gemm_block16w_t<true>(Ah, Al, a_stride, ksteps, wp, NT, tile0, lane, acc);
    else
        gemm_block16w_t<false>(Ah, Al, a_stride, ksteps, wp, NT, tile0, lane, acc);
}

__device__ __forceinline__ int wide16w_tiles_per_wave(int n_tiles) { return n_tiles > 8 ? 2 : 1; }
__device__ __forceinline__ int wide16w_tile0(int n_tiles, int wave) { return n_tiles > 8 ? 2 * wave : wave; }

__device__ __forceinline__ void layer_gemm16w(const nefii_layer &L, const Lds16w &lds, int n_tiles, f32x16 (&acc)[4],
                                              int &nct) {
    // readfirstlane: tell the compiler the wave index is wave-uniform (scalar branches, no exec masking of MFMAs)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // column tiles of this wave: a pair (tiles 2w, 2w+1) for layers wider than 8 tiles, ONE tile (w) for narrower ones -
    // a 256-wide net (conf_neus.conf) then keeps all 8 waves on the matrix cores instead of 4
    const int tile0 = wide16w_tile0(n_tiles, wave);
    nct = n_tiles - tile0;
    nct = nct < 0 ? 0 : (nct > wide16w_tiles_per_wave(n_tiles) ? wide16w_tiles_per_wave(n_tiles) : nct);
    zero_acc(acc);
    const half8 *wp = reinterpret_cast<const half8 *>(L.w_f16x3);
    gemm_block16w(lds.Xh, lds.Xl, XS16, L.k_x >> 4, wp, n_tiles, tile0, lane, nct, acc);
    gemm_block16w(lds.Eh, lds.El, ES16, L.k_e >> 4, wp + (size_t)(L.k_x >> 4) * n_tiles * 2 * 64, n_tiles, tile0, lane,
                  nct, acc);
}

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

// 16 * Softplus(beta=100)(v) from zs = 16 v (the A16_SCALE'd value), for the tracer's split-precision epilogue:
//   softplus(v) = (max(z,0) + log1p(exp(-|z|)))/100, z = 100 v; exp2/log2 on the transcendental units.
// Absolute error ~1e-9 (the 1+e rounding only matters where the result is ~1e-7), which is far below the fp32
// round-off of the 512-term dot products around it; 5 VALU + 2 transcendental instructions per value.
__device__ __forceinline__ float softplus100_s16(float zs) {
    const float C_T = -1.44269504088896340736f * 100.f / A16_SCALE;     // -(log2 e) * 100/16
    const float C_L = 0.69314718055994530942f * A16_SCALE / 100.f;      // ln2 * 16/100
    const float t = __builtin_fabsf(zs) * C_T;
    const float u = 1.f + __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(__builtin_amdgcn_logf(u), C_L, __builtin_fmaxf(zs, 0.f));
}

// The same for the single-pass (coarse) evaluator, two values per instruction in packed fp16: its results are rounded to
// fp16 anyway, and what the coarse pass may get wrong is measured per network (nefii_tracer_params.coarse_tau), so the
// epilogue - a quarter of that evaluator's tile time, and paced by the quarter-rate transcendental unit - can afford fp16
// arithmetic behind the fp32 bias add, and ONE transcendental per value instead of two:
//   16 softplus = max(zs, 0) + (16 ln2 / 100) log2(1 + e),  e = 2^(-|zs| 100 log2(e) / 16) in (0, 1],
//   log2(1 + e) ~ e (a1 + e (a2 + e (a3 + e a4)))  (minimax on [0, 1]: 1.0e-4; with the fp16 Horner steps the result is
//   within 3.1e-4 of the exact value, the v_exp / v_add / v_log chain it replaces within 3.2e-4).
// |z| (v_and), x C_T (v_pk_mul), 2 x v_exp_f16, 4 x v_pk_fma (the last one adds max(z, 0), v_pk_max).
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2v softplus100_s16_pk(half2v zs) {
    const _Float16 C_T = (_Float16)(-1.44269504088896340736f * 100.f / A16_SCALE);
    constexpr float C_L = 0.69314718055994530942f * A16_SCALE / 100.f;
    const _Float16 B1 = (_Float16)(1.43901483f * C_L), B2 = (_Float16)(-0.67994519f * C_L);
    const _Float16 B3 = (_Float16)(0.32559803f * C_L), B4 = (_Float16)(-0.08477006f * C_L);
    const half2v t = __builtin_elementwise_abs(zs) * C_T;
    half2v e;
    e[0] = __builtin_exp2f16(t[0]);
    e[1] = __builtin_exp2f16(t[1]);
    half2v q = __builtin_elementwise_fma(e, half2v{B4, B4}, half2v{B3, B3});
    q = __builtin_elementwise_fma(e, q, half2v{B2, B2});
    q = __builtin_elementwise_fma(e, q, half2v{B1, B1});
    return __builtin_elementwise_fma(e, q, __builtin_elementwise_max(zs, half2v{(_Float16)0.f, (_Float16)0.f}));
}
// four accumulator values -> bias, activation, packed halves
__device__ __forceinline__ half4 softplus100_s16_pk4(const float4v &av, float k16, const float4v &bs) {
    typedef float float2v __attribute__((ext_vector_type(2)));
    const float2v z01 = {__builtin_fmaf(av[0], k16, bs[0]), __builtin_fmaf(av[1], k16, bs[1])};
    const float2v z23 = {__builtin_fmaf(av[2], k16, bs[2]), __builtin_fmaf(av[3], k16, bs[3])};
    const half2v r01 = softplus100_s16_pk(__builtin_convertvector(z01, half2v));
    const half2v r23 = softplus100_s16_pk(__builtin_convertvector(z23, half2v));
    return half4{r01[0], r01[1], r23[0], r23[1]};
}

// transposed accumulator walk of the wide kernel: BODY sees `query` (row of the tile), `f0` (first of 4 consecutive
// features) and `v` (float4v: the 4 accumulator values)
#define NEFII_FOR_ACC_WT(acc, nct, n_tiles, BODY)                                         \
    {                                                                                     \
        const int _wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), _lane = threadIdx.x & 63; \
        _Pragma("unroll") for (int _c = 0; _c < 2; ++_c) if (_c < (nct)) {                \
            _Pragma("unroll") for (int _g = 0; _g < 4; ++_g) {                            \
                const int f0 = 32 * (wide16w_tile0((n_tiles), _wave) + _c) + 8 * _g + 4 * (_lane >> 5); \
                _Pragma("unroll") for (int _rt = 0; _rt < 2; ++_rt) {                     \
                    const int query = 32 * _rt + (_lane & 31);                            \
                    const f32x16 &_a = (acc)[_rt * 2 + _c];                               \
                    const float4v v = {_a[4 * _g], _a[4 * _g + 1], _a[4 * _g + 2], _a[4 * _g + 3]}; \
                    BODY                                                                  \
                }                                                                         \
            }                                                                             \
        }                                                                                 \
    }

// ================================================================================================
// Pipelined wide variant ("16p"): the arithmetic of the wide kernel, restructured around what bounds it - the
// ~70 GB/s per CU at which weight fragments arrive from L2 (MI355X_MICROARCH.md, "Indexed rows"): a 64-query tile
// pulls the whole 7.3 MB network through that pipe, which takes as long as its matrix-core work.  Both only overlap
// if the fragment stream never drains and runs far enough ahead of the MFMAs that consume it:
//   * ONE activation image per half (hi / lo), 584 halves per row = hidden columns [0,512) | encoding [512,576):
//     a layer's input is the contiguous column range [512 - k_x, 512 + k_e), its k-loop one uniform run;
//   * the hidden layers' fragments re-packed as ONE stream per wave (nefii_pack_sdf_stream: layers back to back,
//     wrapping to the next tile's first layer), read by a free-running cursor: the prefetch distance (NB-1 k-steps)
//     does not know about layer boundaries;
//   * NB statically named register stages; with 8 stages a layer starts at stage 0 or 4 (all k-step counts are
//     multiples of 4), hence two instantiations of the k-loop;
//   * no conditional loads: every k-step issues exactly four 1-KiB loads, so the s_waitcnt vmcnt() the compiler
//     derives is static ("all but the youngest NB-1 stages") instead of a vmcnt(0) drain at every join - sched_barriers keep
//     its scheduler from sinking the prefetches back to their uses;
//   * a layer's 64 biases per wave travel in ONE register (lane j holds bias[64 wave + j], fetched at the top of the
//     layer, older than all its prefetches) and reach their lanes through ds_bpermute in the epilogue.
// Shapes: every hidden layer 512 wide (n_pad), k_x in {0,512}, k_e in {0,64}, 512-deep last layer - fits16p();
// other nets take the generic wide kernel.
// ================================================================================================
constexpr int XP16 = 584;               // halves per row (1168 B = 16 B * 73: ds_read_b128 stays conflict-free)
constexpr int EP16 = 512;               // first encoding column
struct Lds16p {
    _Float16 Xh[TILE_W * XP16], Xl[TILE_W * XP16];
    _Float16 tail[64];                  // the A-fragment prefetch of a layer's (non-existent) next k-step lands here
};
// Workgroup shape: NW = 8 waves (two per SIMD, 256 registers each, 2 column tiles per wave, 4 stages = 96 KiB of
// fragments in flight per CU).  The 4-wave shape (one wave per SIMD, 512 registers, 8 stages) is written out below but
// not instantiated: hipcc 7.2 spills ~540 registers on it.
template <int NW>
struct P16 {
    static constexpr int NC = 16 / NW;              // column tiles per wave
    static constexpr int NB = NW == 4 ? 8 : 4;      // fragment stages (6 with two waves per SIMD: hipcc spills)
    static constexpr int STEP = 2 * NC * 64;        // half8 per k-step of one wave's stream
    struct Stage {
        half8 f[2 * NC];                            // hi, lo of each column tile
    };
};
// free-running read position in the wave's fragment stream
struct PCursor {
    const half8 *base;                  // lane's pointer to k-step 0
    unsigned off, bytes;                // current byte offset, stream length
};

template <int NW>
__device__ __forceinline__ void pload(typename P16<NW>::Stage &st, PCursor &c) {
    const half8 *p = reinterpret_cast<const half8 *>(reinterpret_cast<const char *>(c.base) + c.off);
#pragma unroll
    for (int i = 0; i < 2 * P16<NW>::NC; ++i) st.f[i] = p[64 * i];
    c.off += P16<NW>::STEP * 16;
    c.off = c.off == c.bytes ? 0u : c.off;
}

template <int RT>
__device__ __forceinline__ void pload_a(AStage16w &st, const _Float16 *ah, const _Float16 *al, int s) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        st.ah[rt] = *reinterpret_cast<const half8 *>(ah + rt * 32 * XP16 + 16 * s);
        st.al[rt] = *reinterpret_cast<const half8 *>(al + rt * 32 * XP16 + 16 * s);
    }
}

template <int NW, int RT>
__device__ __forceinline__ void pmfma(const AStage16w &sa, const typename P16<NW>::Stage &sb,
                                      f32x16 (&acc)[RT * P16<NW>::NC]) {
    constexpr int NC = P16<NW>::NC;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x16 &a = acc[rt * NC + c];
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(sb.f[2 * c], sa.ah[rt], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(sb.f[2 * c + 1], sa.ah[rt], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(sb.f[2 * c], sa.al[rt], a, 0, 0, 0);
        }
}

// one k-step: prefetch the fragments NB-1 steps ahead into the stage consumed last step, read the next step's
// activation fragments, multiply this step's.  J = stage of this step, U = its parity (activation double buffer).
template <int NW, int RT, int J, int U>
__device__ __forceinline__ void pstep(typename P16<NW>::Stage (&b)[P16<NW>::NB], AStage16w (&a)[2], PCursor &cur,
                                      const _Float16 *ah, const _Float16 *al, int s,
                                      f32x16 (&acc)[RT * P16<NW>::NC]) {
    constexpr int NB = P16<NW>::NB;
    // Issue order inside the step: the MFMAs lead and the step's memory instructions (prefetch of the stage consumed last
    // step, next step's activation fragments) are spread between them, two MFMAs per instruction.  Issuing the eight
    // memory instructions in a block ahead of the MFMAs leaves the matrix pipe idle while they issue and cost 13 % of
    // the tile time (s_memtime stamps, tools/stamps.py); the closing sched_barrier keeps the compiler from sinking
    // a prefetch past the step (it otherwise moves every load down to its use and drains the pipeline).
    pload<NW>(b[(J + NB - 1) % NB], cur);
    pload_a<RT>(a[(U + 1) & 1], ah, al, s + 1);
    pmfma<NW, RT>(a[U & 1], b[J], acc);
#pragma unroll
    for (int i = 0; i < 2 * P16<NW>::NC; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // 2 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // 1 VMEM read
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, P16<NW>::NC, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 DS read
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NW, int RT, int J0>
__device__ __forceinline__ void pstep4(typename P16<NW>::Stage (&b)[P16<NW>::NB], AStage16w (&a)[2], PCursor &cur,
                                       const _Float16 *ah, const _Float16 *al, int s,
                                       f32x16 (&acc)[RT * P16<NW>::NC]) {
    pstep<NW, RT, J0, 0>(b, a, cur, ah, al, s, acc);
    pstep<NW, RT, J0 + 1, 1>(b, a, cur, ah, al, s + 1, acc);
    pstep<NW, RT, J0 + 2, 0>(b, a, cur, ah, al, s + 2, acc);
    pstep<NW, RT, J0 + 3, 1>(b, a, cur, ah, al, s + 3, acc);
}

// k-loop of one layer whose first k-step sits in stage PH; ks is a multiple of 4 (PH is 0, or 4 with 8 stages)
template <int NW, int RT, int PH>
__device__ __forceinline__ void pgemm(int ks, typename P16<NW>::Stage (&b)[P16<NW>::NB], AStage16w (&a)[2],
                                      PCursor &cur, const _Float16 *ah, const _Float16 *al,
                                      f32x16 (&acc)[RT * P16<NW>::NC]) {
    constexpr int NB = P16<NW>::NB;
    static_assert((NB == 4 && PH == 0) || (NB == 8 && (PH == 0 || PH == 4)), "stage bookkeeping");
    int s = 0;
    if constexpr (NB == 4) {
        for (; s < ks; s += 4) pstep4<NW, RT, 0>(b, a, cur, ah, al, s, acc);
    } else {
        if (PH == 4) {
            pstep4<NW, RT, 4>(b, a, cur, ah, al, 0, acc);
            s = 4;
        }
        for (; s + 8 <= ks; s += 8) {
            pstep4<NW, RT, 0>(b, a, cur, ah, al, s, acc);
            pstep4<NW, RT, 4>(b, a, cur, ah, al, s + 4, acc);
        }
        if (s < ks) pstep4<NW, RT, 0>(b, a, cur, ah, al, s, acc);
    }
}

template <int NW>
__device__ __forceinline__ void encode_tile16p(const nefii_mlp &m, const float *raw, Lds16p &lds, int k_e) {
    const int tid = threadIdx.x;
    const int p = tid & 63, part = tid >> 6;
    const int w0 = enc_width(m.enc_freqs[0]);
    for (int c = part; c < k_e; c += NW) {
        const float val = c < w0 ? enc_value(raw + p * 9, c) : 0.f;
        split16a(val, lds.Xh[p * XP16 + EP16 + c], lds.Xl[p * XP16 + EP16 + c]);
    }
}

// start of a workgroup: stages 0..NB-2 <- k-steps 0..NB-2 of the stream
template <int NW>
__device__ __forceinline__ void prime16p(const nefii_mlp &m, typename P16<NW>::Stage (&b)[P16<NW>::NB],
                                         PCursor &cur) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int G = 0;
    for (int l = 0; l < m.n_layers - 1; ++l) G += (m.layer[l].k_x + m.layer[l].k_e) >> 4;
    cur.bytes = (unsigned)G * P16<NW>::STEP * 16;
    cur.base = reinterpret_cast<const half8 *>(m.w_stream) + (size_t)wave * G * P16<NW>::STEP + lane;
    cur.off = 0;
#pragma unroll
    for (int u = 0; u < P16<NW>::NB - 1; ++u) pload<NW>(b[u], cur);
}

// One tile of 32 * RT queries through the whole SDF network (RT = 2: the 64-query tile; RT = 1: 32 queries, half the
// matrix work and epilogue on the same fragment stream - for rounds with fewer 64-query tiles than half the CUs).
// One 64-query tile through the whole SDF network.  Stage/cursor state runs on from tile to tile (ph = stage of the
// next k-step).  raw[64][9]: the points (overwritten with the last layer's partial sums); dest[64]: where each SDF
// value goes (nullptr = padding row).
// epilogue arithmetic of one wave's 64 x 32*NC block: bias, activation, (16 x) hi/lo split, packed four features at a
// time (transposed accumulator: lane = query, registers = 4 consecutive features x 4 groups).  FAST = Softplus(beta 100).
template <int NW, int RT, bool FAST>
__device__ __forceinline__ void pepilogue(const f32x16 (&acc)[RT * P16<NW>::NC], const float (&bvec)[P16<NW>::NC / 2 + 1],
                                          float k16, int h, int act, half4 (&phi)[P16<NW>::NC * 4 * RT],
                                          half4 (&plo)[P16<NW>::NC * 4 * RT]) {
    constexpr int NC = P16<NW>::NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int bsrc = __builtin_bit_cast(int, bvec[c >> 1] * A16_SCALE);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4v bs;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                bs[k] = __builtin_bit_cast(float,
                                           __builtin_amdgcn_ds_bpermute(4 * (32 * (c & 1) + 8 * g + 4 * h + k), bsrc));
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const f32x16 &av = acc[rt * NC + c];
                float4v hs;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float zs = __builtin_fmaf(av[4 * g + k], k16, bs[k]);
                    hs[k] = FAST ? softplus100_s16(zs) : act_fwd(zs * (1.f / A16_SCALE), act) * A16_SCALE;
                }
                const half4 hi = __builtin_convertvector(hs, half4);
                phi[(c * 4 + g) * RT + rt] = hi;
                plo[(c * 4 + g) * RT + rt] = __builtin_convertvector(hs - __builtin_convertvector(hi, float4v), half4);
            }
        }
    }
}

#ifdef NEFII_STAMPS     /* timing instrumentation: s_memtime at 5 points per layer, workgroup 0, into g_stamps */
__device__ unsigned long long g_stamps[2 * 8 * 12 * 5];
__device__ int g_stamp_tile;
#define NEFII_STAMP(i)                                                                                   \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && g_stamp_tile < 2)                                  \
        g_stamps[((g_stamp_tile * 8 + wave) * 12 + l) * 5 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define NEFII_STAMP(i)
#endif

template <int NW, int RT>
__device__ __forceinline__ void sdf_tile16p(const nefii_mlp &m, Lds16p &lds, float *raw, float *const *dest,
                                            typename P16<NW>::Stage (&b)[P16<NW>::NB], PCursor &cur, int &ph, int ke) {
    constexpr int NC = P16<NW>::NC, NB = P16<NW>::NB;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int NH = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    const float k16 = inv_scale * A16_SCALE;
    encode_tile16p<NW>(m, raw, lds, ke);
    __syncthreads();
    const _Float16 *ah0 = lds.Xh + r * XP16 + 8 * h, *al0 = lds.Xl + r * XP16 + 8 * h;
    for (int l = 0; l < NH; ++l) {
        const nefii_layer &L = m.layer[l];
        const int ks = (L.k_x + L.k_e) >> 4;
        const _Float16 *ah = ah0 + (EP16 - L.k_x), *al = al0 + (EP16 - L.k_x);
        const float *bp = L.bias + 32 * NC * wave + lane;
        // everything prefetched so far has had a whole epilogue to land: an explicit drain here costs nothing and gives
        // the compiler's waitcnt pass a known state at the loop headers
        asm volatile("" ::"s"(ks), "v"(ah), "v"(al), "v"(bp));
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        NEFII_STAMP(0);
        float bvec[NC / 2 + 1];
#pragma unroll
        for (int i = 0; i < NC / 2; ++i) bvec[i] = bp[64 * i];
        f32x16 acc[RT * NC];
#pragma unroll
        for (int j = 0; j < RT * NC; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        AStage16w a[2];
        pload_a<RT>(a[0], ah, al, 0);
        if (NB == 4 || ph == 0)
            pgemm<NW, RT, 0>(ks, b, a, cur, ah, al, acc);
        else
            pgemm<NW, RT, (NB == 8 ? 4 : 0)>(ks, b, a, cur, ah, al, acc);
        ph = (ph + ks) % NB;
        NEFII_STAMP(1);
        // epilogue, arithmetic first: activation and hi/lo split of the wave's 64 x 64 block, packed in registers (the
        // accumulators' own count).  It needs nothing the other waves still read, so it runs BEFORE the barrier: the
        // SIMD partner that lost the matrix pipe arbitration is still in its k-loop then (MFMA beside VALU), see
        // tools/stamps.py.  Only the stores into the activation image wait for everyone to be done reading it.
        half4 phi[NC * 4 * RT], plo[NC * 4 * RT];
        // the activation switch sits OUTSIDE the 64-value loop: with the test inside it every value went through its own
        // scalar branch, the compiler could not interleave the values' exp -> add -> log -> fma chains, and the epilogue
        // ran at the latency of one chain per value (~120 cycles) instead of the VALU's throughput
        if (m.act == NEFII_ACT_SOFTPLUS100)
            pepilogue<NW, RT, true>(acc, bvec, k16, h, m.act, phi, plo);
        else
            pepilogue<NW, RT, false>(acc, bvec, k16, h, m.act, phi, plo);
        __syncthreads();
        NEFII_STAMP(2);
        _Float16 *xh = lds.Xh + (EP16 - L.n_pad), *xl = lds.Xl + (EP16 - L.n_pad);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f0 = 32 * (NC * wave + c) + 8 * g + 4 * h;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int query = 32 * rt + r;
                    *reinterpret_cast<half4 *>(xh + query * XP16 + f0) = phi[(c * 4 + g) * RT + rt];
                    *reinterpret_cast<half4 *>(xl + query * XP16 + f0) = plo[(c * 4 + g) * RT + rt];
                }
            }
        NEFII_STAMP(3);
        __syncthreads();
        NEFII_STAMP(4);
    }
    // last layer, column 0 only (the SDF value): its 512-deep dot product is split over the waves, partial sums
    // meet in LDS
    {
        const nefii_layer &L = m.layer[NH];
        const int NT = L.n_pad >> 5;
        const half8 *wl = reinterpret_cast<const half8 *>(L.w_f16x3) + lane;
        const _Float16 *ah = ah0 + (EP16 - L.k_x), *al = al0 + (EP16 - L.k_x);
        const int ksw = (L.k_x >> 4) / NW;          // k-steps per wave
        f32x16 acc2[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[rt][i] = 0.f;
        for (int u = 0; u < ksw; ++u) {
            const int s = wave * ksw + u;
            const half8 wh = wl[(size_t)s * NT * 128], wlo = wl[(size_t)s * NT * 128 + 64];
            AStage16w a;
            pload_a<RT>(a, ah, al, s);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, a.ah[rt], acc2[rt], 0, 0, 0);
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, a.ah[rt], acc2[rt], 0, 0, 0);
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, a.al[rt], acc2[rt], 0, 0, 0);
            }
        }
        if (h == 0) {       // feature 0 = register 0 of lanes 0..31
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) raw[wave * TILE_W + 32 * rt + r] = acc2[rt][0];
        }
        __syncthreads();
        if (threadIdx.x < 32 * RT) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += raw[w * TILE_W + threadIdx.x];
            float *d = dest[threadIdx.x];
            if (d) *d = sum * inv_scale + L.bias[0];
        }
        __syncthreads();
    }
}

// ================================================================================================
// "16q": the pipelined evaluator on v_mfma_f32_16x16x32_f16 (nefii_mlp.reserved == 1 selects it and the matching stream
// layout).  Same tile, stream cursor, stages, LDS image and epilogue idea as "16p"; what changes is the matrix
// instruction: 16 x 16 output tiles, 32-deep k-steps.  A wave still owns 64 features (4 feature tiles) x all queries.
// The stream's unit is a HALF step: the hi/lo fragments of two feature tiles for one 32-deep k-step (4 KiB, as
// before), consumed by 2 x QT x 3 MFMAs; the activation fragments (QT query tiles, hi and lo) are read once per full
// step.  Weights are the A operand (rows = features), activations the B operand (columns = queries): the accumulator
// holds features 4 (lane>>4) + 0..3 of query lane&15.
// ================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
// FT = 16-feature tiles per wave: 4 for 512-wide hidden layers, 2 for 256-wide ones (conf_neus.conf).  With FT = 2 a
// stream unit is a whole 32-deep k-step (both feature tiles), every layer's K is padded to a multiple of 128 with zero
// weights (unit counts stay multiples of 4: one stage phase) and a tile holds up to 96 queries (QT = 6).
template <int FT>
struct QGeo {
    static constexpr int HW = 128 * FT;                 // hidden width = first encoding column
    static constexpr int EW = FT == 4 ? 64 : 128;       // encoding columns incl. zero padding
#ifndef NEFII_XPAD
#define NEFII_XPAD 8
#endif
    static constexpr int XP = HW + EW + NEFII_XPAD;     // halves per row: 584 / 392 (16 B * odd)
    static constexpr int ROWS = FT == 4 ? 64 : 96;      // queries per tile at most
};
template <int FT>
struct LdsQ {
    _Float16 Xh[QGeo<FT>::ROWS * QGeo<FT>::XP], Xl[QGeo<FT>::ROWS * QGeo<FT>::XP];
    _Float16 tail[64];
};
// k-loop units of a layer: half steps (16 deep) for FT = 4, whole 32-deep steps of the 128-padded K for FT = 2
template <int FT>
__host__ __device__ __forceinline__ int q_units(const nefii_layer &L) {
    return FT == 4 ? (L.k_x + L.k_e) >> 4 : ((L.k_x + L.k_e + 127) & ~127) >> 5;
}
template <int QT>
struct QAct {
    half8 h[QT], l[QT];
};

template <int QT, int XP>
__device__ __forceinline__ void qload_a(QAct<QT> &st, const _Float16 *ah, const _Float16 *al, int s32) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        st.h[qt] = *reinterpret_cast<const half8 *>(ah + qt * 16 * XP + 32 * s32);
        st.l[qt] = *reinterpret_cast<const half8 *>(al + qt * 16 * XP + 32 * s32);
    }
}

// one half step: HALF selects the feature-tile pair (ft = 2 HALF, 2 HALF + 1) whose fragments stage J holds
template <int QT, int FT, int J, int HALF, int ABUF, bool LOADA, int XP = QGeo<FT>::XP>
__device__ __forceinline__ void qstep(P16<8>::Stage (&b)[4], QAct<QT> (&a)[2], PCursor &cur, const _Float16 *ah,
                                      const _Float16 *al, int s32, f32x4 (&acc)[FT * QT]) {
    pload<8>(b[(J + 3) % 4], cur);
    if (LOADA) qload_a<QT, XP>(a[ABUF ^ 1], ah, al, s32 + 1);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 &c = acc[(2 * HALF + f) * QT + qt];
#ifdef NEFII_STREAM_ONLY
            asm volatile("" ::"v"(b[J].f[2 * f]), "v"(b[J].f[2 * f + 1]), "v"(a[ABUF].h[qt]), "v"(a[ABUF].l[qt]));
            (void)c;
#else
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[J].f[2 * f], a[ABUF].h[qt], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[J].f[2 * f + 1], a[ABUF].h[qt], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[J].f[2 * f], a[ABUF].l[qt], c, 0, 0, 0);
#endif
        }
    // MFMAs lead, the memory instructions are spread between them (see pstep)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, QT, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    if (LOADA) {
#pragma unroll
        for (int i = 0; i < 2 * QT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// k-loop of one layer: `units` stream units (a multiple of 4), stage 0 first
template <int QT, int FT, int XP = QGeo<FT>::XP>
__device__ __forceinline__ void qgemm(int units, P16<8>::Stage (&b)[4], QAct<QT> (&a)[2], PCursor &cur,
                                      const _Float16 *ah, const _Float16 *al, f32x4 (&acc)[FT * QT]) {
    if constexpr (FT == 4) {
        for (int hs = 0; hs < units; hs += 4) {
            const int s32 = hs >> 1;
            qstep<QT, FT, 0, 0, 0, true, XP>(b, a, cur, ah, al, s32, acc);
            qstep<QT, FT, 1, 1, 0, false, XP>(b, a, cur, ah, al, s32, acc);
            qstep<QT, FT, 2, 0, 1, true, XP>(b, a, cur, ah, al, s32 + 1, acc);
            qstep<QT, FT, 3, 1, 1, false, XP>(b, a, cur, ah, al, s32 + 1, acc);
        }
    } else {
        for (int s32 = 0; s32 < units; s32 += 4) {
            qstep<QT, FT, 0, 0, 0, true>(b, a, cur, ah, al, s32, acc);
            qstep<QT, FT, 1, 0, 1, true>(b, a, cur, ah, al, s32 + 1, acc);
            qstep<QT, FT, 2, 0, 0, true>(b, a, cur, ah, al, s32 + 2, acc);
            qstep<QT, FT, 3, 0, 1, true>(b, a, cur, ah, al, s32 + 3, acc);
        }
    }
}

// ---- deep-prefetch variant for the 32-query instance (FT = 4, QT = 2) ---------------------------------------------------
// A 32-query tile has half the matrix work of a 64-query one but pulls the same 7.6 MB of fragments through its CU: with
// 3 units (12 KB per wave) in flight it is bound by the latency of that stream (~73 GB/s per CU however many CUs run),
// ~105 us against ~55 us of MFMA time.  The small instance has the registers for 8 stages (7 units = 28 KB per wave in
// flight); its stream is a second copy with every layer's K zero-padded to a multiple of 128 (unit counts = multiples
// of the 8 stages, one loop body).
template <int QT, int J, int HALF, int ABUF, bool LOADA>
__device__ __forceinline__ void qstep8(P16<8>::Stage (&b)[8], QAct<QT> (&a)[2], PCursor &cur, const _Float16 *ah,
                                       const _Float16 *al, int s32, f32x4 (&acc)[4 * QT]) {
    pload<8>(b[(J + 7) % 8], cur);
    if (LOADA) qload_a<QT, QGeo<4>::XP>(a[ABUF ^ 1], ah, al, s32 + 1);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 &c = acc[(2 * HALF + f) * QT + qt];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[J].f[2 * f], a[ABUF].h[qt], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[J].f[2 * f + 1], a[ABUF].h[qt], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[J].f[2 * f], a[ABUF].l[qt], c, 0, 0, 0);
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, QT, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    if (LOADA) {
#pragma unroll
        for (int i = 0; i < 2 * QT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// k-loop of one layer: `units` half steps, a multiple of 8, stage 0 first
template <int QT>
__device__ __forceinline__ void qgemm8(int units, P16<8>::Stage (&b)[8], QAct<QT> (&a)[2], PCursor &cur,
                                       const _Float16 *ah, const _Float16 *al, f32x4 (&acc)[4 * QT]) {
    for (int hs = 0; hs < units; hs += 8) {
        const int s32 = hs >> 1;
        qstep8<QT, 0, 0, 0, true>(b, a, cur, ah, al, s32, acc);
        qstep8<QT, 1, 1, 0, false>(b, a, cur, ah, al, s32, acc);
        qstep8<QT, 2, 0, 1, true>(b, a, cur, ah, al, s32 + 1, acc);
        qstep8<QT, 3, 1, 1, false>(b, a, cur, ah, al, s32 + 1, acc);
        qstep8<QT, 4, 0, 0, true>(b, a, cur, ah, al, s32 + 2, acc);
        qstep8<QT, 5, 1, 0, false>(b, a, cur, ah, al, s32 + 2, acc);
        qstep8<QT, 6, 0, 1, true>(b, a, cur, ah, al, s32 + 3, acc);
        qstep8<QT, 7, 1, 1, false>(b, a, cur, ah, al, s32 + 3, acc);
    }
}

// half steps of a layer in the padded stream
__host__ __device__ __forceinline__ int q_units8(const nefii_layer &L) { return ((L.k_x + L.k_e + 127) & ~127) >> 4; }

template <int QT, int FT, bool FAST>
__device__ __forceinline__ void qepilogue(const f32x4 (&acc)[FT * QT], float bvec, float k16, int lane, int act,
                                          half4 (&phi)[FT * QT], half4 (&plo)[FT * QT]) {
    const int bsrc = __builtin_bit_cast(int, bvec * A16_SCALE);
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        float4v bs;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            bs[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (16 * ft + 4 * (lane >> 4) + k), bsrc));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const f32x4 &av = acc[ft * QT + qt];
            float4v hs;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float zs = __builtin_fmaf(av[k], k16, bs[k]);
                hs[k] = FAST ? softplus100_s16(zs) : act_fwd(zs * (1.f / A16_SCALE), act) * A16_SCALE;
            }
            const half4 hi = __builtin_convertvector(hs, half4);
            phi[ft * QT + qt] = hi;
            plo[ft * QT + qt] = __builtin_convertvector(hs - __builtin_convertvector(hi, float4v), half4);
        }
    }
}

template <int FT>
__device__ __forceinline__ void encode_tile16q(const nefii_mlp &m, const float *raw, LdsQ<FT> &lds, int rows) {
    constexpr int XP = QGeo<FT>::XP, EP = QGeo<FT>::HW, EW = QGeo<FT>::EW;
    const int w0 = enc_width(m.enc_freqs[0]);
    for (int i = threadIdx.x; i < rows * EW; i += 512) {
        const int p = i / EW, c = i - p * EW;
        const float val = c < w0 ? enc_value(raw + p * 9, c) : 0.f;
        split16a(val, lds.Xh[p * XP + EP + c], lds.Xl[p * XP + EP + c]);
    }
}

// stages 0..2 <- units 0..2 of the stream (start of a workgroup)
template <int FT, int NB = 4>
__device__ __forceinline__ void prime16q(const nefii_mlp &m, P16<8>::Stage (&b)[NB], PCursor &cur) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int G = 0, G8 = 0;
    for (int l = 0; l < m.n_layers - 1; ++l) G += q_units<FT>(m.layer[l]), G8 += q_units8(m.layer[l]);
    const half8 *base = reinterpret_cast<const half8 *>(m.w_stream);
    if (NB == 8) base += (size_t)8 * G * 256, G = G8;       // the padded copy follows the plain stream
    cur.bytes = (unsigned)G * 4096;
    cur.base = base + (size_t)wave * G * 256 + lane;
    cur.off = 0;
#pragma unroll
    for (int u = 0; u < NB - 1; ++u) pload<8>(b[u], cur);
}

// One tile of 16 * QT queries through the whole SDF network (FT = 4: QT = 4 / 2 for 64 / 32 queries; FT = 2: QT = 6 / 2).
// coarse_old / audit (the tracer's refined coarse samples only): coarse_old[query] = the value the single-pass evaluator gave
// this query (NaN: none) - the largest |coarse - split| of the tile goes to *audit (atomic max on the float's bits)
template <int QT, int FT, int NB = 4>
__device__ __forceinline__ void sdf_tile16q(const nefii_mlp &m, LdsQ<FT> &lds, float *raw, float *const *dest,
                                            P16<8>::Stage (&b)[NB], PCursor &cur, const float *coarse_old = nullptr,
                                            int *audit = nullptr) {
    static_assert(NB == 4 || FT == 4, "the deep-prefetch variant is the 512-wide shape's");
    constexpr int NW = 8, RT = QT / 2, XP = QGeo<FT>::XP, EP = QGeo<FT>::HW, RMAX = QGeo<FT>::ROWS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NH = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    const float k16 = inv_scale * A16_SCALE;
    // a layer's bias is fetched one layer ahead (behind the previous k-loop): read at the top of its own layer it would
    // sit behind the next layer's prefetched fragments in vmcnt order and the epilogue would open with a vmcnt(0)
    const int boff = 16 * FT * wave + (lane & (16 * FT - 1));
    float bnext = m.layer[0].bias[boff];
    encode_tile16q<FT>(m, raw, lds, 16 * QT);
    __syncthreads();
    const _Float16 *qh0 = lds.Xh + (lane & 15) * XP + 8 * (lane >> 4), *ql0 = lds.Xl + (lane & 15) * XP + 8 * (lane >> 4);
    for (int l = 0; l < NH; ++l) {
        const nefii_layer &L = m.layer[l];
        const int units = NB == 8 ? q_units8(L) : q_units<FT>(L);
        const _Float16 *ah = qh0 + (EP - L.k_x), *al = ql0 + (EP - L.k_x);
        const float *bp = m.layer[l + 1 < NH ? l + 1 : l].bias + boff;
        asm volatile("" ::"s"(units), "v"(ah), "v"(al), "v"(bp));
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): known state for the waitcnt pass (see 16p)
        __builtin_amdgcn_sched_barrier(0);
        const float bvec = bnext;
        f32x4 acc[FT * QT];
#pragma unroll
        for (int j = 0; j < FT * QT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
        QAct<QT> a[2];
        qload_a<QT, XP>(a[0], ah, al, 0);
        if constexpr (NB == 8)
            qgemm8<QT>(units, b, a, cur, ah, al, acc);
        else
            qgemm<QT, FT>(units, b, a, cur, ah, al, acc);
        bnext = *bp;
        __builtin_amdgcn_sched_barrier(0);
        half4 phi[FT * QT], plo[FT * QT];
        if (m.act == NEFII_ACT_SOFTPLUS100)
            qepilogue<QT, FT, true>(acc, bvec, k16, lane, m.act, phi, plo);
        else
            qepilogue<QT, FT, false>(acc, bvec, k16, lane, m.act, phi, plo);
        __syncthreads();
        _Float16 *xh = lds.Xh + (EP - L.n_pad), *xl = lds.Xl + (EP - L.n_pad);
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
            const int f0 = 16 * FT * wave + 16 * ft + 4 * (lane >> 4);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int query = 16 * qt + (lane & 15);
                *reinterpret_cast<half4 *>(xh + query * XP + f0) = phi[ft * QT + qt];
                *reinterpret_cast<half4 *>(xl + query * XP + f0) = plo[ft * QT + qt];
            }
        }
        __syncthreads();
    }
    // last layer, column 0 only: 32x32x16 fragments of the layer's own w_f16x3, K split over the waves (as in 16p)
    {
        const int r = lane & 31, h = lane >> 5;
        const nefii_layer &L = m.layer[NH];
        const int NT = L.n_pad >> 5;
        const half8 *wl = reinterpret_cast<const half8 *>(L.w_f16x3) + lane;
        const _Float16 *ah = lds.Xh + r * XP + 8 * h + (EP - L.k_x), *al = lds.Xl + r * XP + 8 * h + (EP - L.k_x);
        const int ksw = (L.k_x >> 4) / NW;
        f32x16 acc2[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[rt][i] = 0.f;
        for (int u = 0; u < ksw; ++u) {
            const int s = wave * ksw + u;
            const half8 wh = wl[(size_t)s * NT * 128], wlo = wl[(size_t)s * NT * 128 + 64];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const half8 xh8 = *reinterpret_cast<const half8 *>(ah + rt * 32 * XP + 16 * s);
                const half8 xl8 = *reinterpret_cast<const half8 *>(al + rt * 32 * XP + 16 * s);
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh8, acc2[rt], 0, 0, 0);
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xh8, acc2[rt], 0, 0, 0);
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl8, acc2[rt], 0, 0, 0);
            }
        }
        if (h == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) raw[wave * RMAX + 32 * rt + r] = acc2[rt][0];
        }
        __syncthreads();
        float dm = 0.f;
        if (threadIdx.x < 32 * RT) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += raw[w * RMAX + threadIdx.x];
            float *d = dest[threadIdx.x];
            const float v = sum * inv_scale + L.bias[0];
            if (d) *d = v;
            if (coarse_old) {
                const float o = coarse_old[threadIdx.x];
                if (d && o == o) dm = __builtin_fabsf(o - v);
            }
        }
        if (coarse_old && threadIdx.x < 128) {      // whole waves (32 RT <= 96): every lane takes part in the reduction
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) dm = __builtin_fmaxf(dm, __shfl_xor(dm, s));
            if ((threadIdx.x & 63) == 0 && dm > 0.f) atomicMax(audit, __builtin_bit_cast(int, dm));
        }
        __syncthreads();
    }
}

// ================================================================================================
// "16s": the pipelined evaluator in ONE fp16 pass (hi x hi only, fp32 accumulate) - the tracer's COARSE pass.
// The 100-sample bracket search and the min-SDF search of RayTracing (ray_tracing.py:195-257,309-337; ~85 % of all SDF
// evaluations) only ask for a sign / an argmin: their samples are evaluated here at a third of the split evaluator's
// matrix work and half its fragment stream, and only the samples whose coarse value is within the error bound
// nefii_tracer_params.coarse_tau of a decision (a sign change, the minimum) are re-evaluated in split precision
// (advance_kernel) - the decisions, and with them the tracer's outputs, stay those of the split evaluator.
// Same tile, cursor, 4 register stages and epilogue idea as "16q"; what changes:
//   * the stream unit is one 32-deep k-step of the wave's FT feature tiles, hi fragments only (FT KiB), every layer's
//     K zero-padded to a multiple of 128 (unit counts = multiples of the 4 stages): nefii_pack_sdf_stream's third copy;
//   * the activation image has the hi halves only (16 QT rows x XP halves): with a third of the matrix work per query the
//     tile is bound by its fragment stream unless it holds more queries than the split evaluator's (QT = 6: 96);
//   * FT * QT MFMAs per unit, one activation fragment read per query tile and unit.
// ================================================================================================
template <int FT>
struct SStage {
    half8 f[FT];
};
template <int QT>
struct SAct {
    half8 h[QT];
};
template <int FT, int ROWS>
struct LdsS {
    _Float16 Xh[ROWS * QGeo<FT>::XP];
    _Float16 tail[128];
};
// units (32-deep k-steps of the 128-padded K) of a layer in the single-pass stream
__host__ __device__ __forceinline__ int s_units(const nefii_layer &L) { return ((L.k_x + L.k_e + 127) & ~127) >> 5; }

template <int FT>
__device__ __forceinline__ void sload(SStage<FT> &st, PCursor &c) {
    const half8 *p = reinterpret_cast<const half8 *>(reinterpret_cast<const char *>(c.base) + c.off);
#ifdef NEFII_X_NO_FRAGLOAD  /* experiment (what in the evaluator disturbs its SIMD neighbours) */
    if (c.off == 0xffffffffu)
#endif
#pragma unroll
    for (int i = 0; i < FT; ++i) st.f[i] = p[64 * i];
    c.off += FT * 1024;
    c.off = c.off == c.bytes ? 0u : c.off;
}

template <int QT, int XP>
__device__ __forceinline__ void sload_a(SAct<QT> &st, const _Float16 *ah, int s32) {
#ifdef NEFII_X_NO_ACTLOAD
    if (s32 == 0x7fffffff)
#endif
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) st.h[qt] = *reinterpret_cast<const half8 *>(ah + qt * 16 * XP + 32 * s32);
}

// DB: the next step's activation fragments are read while this step multiplies (two register sets); !DB: one set, read
// at the top of the step that uses it (24-32 registers less for the 96- / 128-query tiles, the SIMD partner covers the wait)
template <int QT, int FT, int J, bool DB>
__device__ __forceinline__ void sstep(SStage<FT> (&b)[4], SAct<QT> (&a)[DB ? 2 : 1], PCursor &cur, const _Float16 *ah, int s32,
                                      f32x4 (&acc)[FT * QT]) {
    sload<FT>(b[(J + 3) % 4], cur);
    if constexpr (DB)
        sload_a<QT, QGeo<FT>::XP>(a[(J + 1) & 1], ah, s32 + 1);
    else
        sload_a<QT, QGeo<FT>::XP>(a[0], ah, s32);
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 &c = acc[ft * QT + qt];
#ifdef NEFII_STREAM_ONLY    /* experiment: the tile without its matrix work - fragments and activation reads stay */
            asm volatile("" ::"v"(b[J].f[ft]), "v"(a[DB ? (J & 1) : 0].h[qt]));
            (void)c;
#else
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[J].f[ft], a[DB ? (J & 1) : 0].h[qt], c, 0, 0, 0);
#endif
        }
    if constexpr (DB) {
        // MFMAs lead; the unit's FT fragment loads and QT activation reads are spread between them (see pstep)
        // (feature tile i: QT MFMAs with its fragment load and its share of the QT reads in their midst)
#define NEFII_SGROUP(i)                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, QT / 2, 0);                                       \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, QT - QT / 2, 0);                                  \
    __builtin_amdgcn_sched_group_barrier(0x100, (QT * ((i) + 1)) / FT - (QT * (i)) / FT, 0);
        NEFII_SGROUP(0)
        NEFII_SGROUP(1)
        if constexpr (FT == 4) {
            NEFII_SGROUP(2)
            NEFII_SGROUP(3)
        }
#undef NEFII_SGROUP
    } else {
        // the step's own activation reads first, then MFMAs with the fragment loads spread between them
        __builtin_amdgcn_sched_group_barrier(0x100, QT, 0);
#pragma unroll
        for (int i = 0; i < FT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, QT / 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, QT - QT / 2, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int QT, int FT, bool DB>
__device__ __forceinline__ void sgemm(int units, SStage<FT> (&b)[4], SAct<QT> (&a)[DB ? 2 : 1], PCursor &cur,
                                      const _Float16 *ah, f32x4 (&acc)[FT * QT]) {
    for (int s = 0; s < units; s += 4) {
        sstep<QT, FT, 0, DB>(b, a, cur, ah, s, acc);
        sstep<QT, FT, 1, DB>(b, a, cur, ah, s + 1, acc);
        sstep<QT, FT, 2, DB>(b, a, cur, ah, s + 2, acc);
        sstep<QT, FT, 3, DB>(b, a, cur, ah, s + 3, acc);
    }
}

template <int QT, int FT, bool FAST>
__device__ __forceinline__ void sepilogue(const f32x4 (&acc)[FT * QT], float bvec, float k16, int lane, int act,
                                          half4 (&phi)[FT * QT]) {
    const int bsrc = __builtin_bit_cast(int, bvec * A16_SCALE);
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        float4v bs;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            bs[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (16 * ft + 4 * (lane >> 4) + k), bsrc));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const f32x4 &av = acc[ft * QT + qt];
            if constexpr (FAST) {
#ifdef NEFII_X_NO_EPILOGUE
                phi[ft * QT + qt] = __builtin_convertvector(av * k16 + bs, half4);
#else
                phi[ft * QT + qt] = softplus100_s16_pk4(av, k16, bs);
#endif
            } else {
                float4v hs;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float zs = __builtin_fmaf(av[k], k16, bs[k]);
                    hs[k] = act_fwd(zs * (1.f / A16_SCALE), act) * A16_SCALE;
                }
                phi[ft * QT + qt] = __builtin_convertvector(hs, half4);
            }
        }
    }
}

template <int FT, int ROWS>
__device__ __forceinline__ void encode_tile16s(const nefii_mlp &m, const float *raw, LdsS<FT, ROWS> &lds, int rows) {
    constexpr int XP = QGeo<FT>::XP, EP = QGeo<FT>::HW, EW = QGeo<FT>::EW;
    const int w0 = enc_width(m.enc_freqs[0]);
    for (int i = threadIdx.x; i < rows * EW; i += 512) {
        const int p = i / EW, c = i - p * EW;
        const float val = c < w0 ? enc_value(raw + p * 9, c) : 0.f;
        lds.Xh[p * XP + EP + c] = (_Float16)(val * A16_SCALE);
    }
}

// units of the split-precision streams that precede the single-pass copy in nefii_mlp.w_stream, in 4 KiB blocks per wave
template <int FT>
__host__ __device__ __forceinline__ void s_stream_geometry(const nefii_mlp &m, int &blocks_before, int &G) {
    int Gq = 0, G8 = 0;
    G = 0;
    for (int l = 0; l < m.n_layers - 1; ++l) {
        Gq += q_units<FT>(m.layer[l]);
        G8 += q_units8(m.layer[l]);
        G += s_units(m.layer[l]);
    }
    blocks_before = Gq + (FT == 4 ? G8 : 0);
}

template <int FT>
__device__ __forceinline__ void prime16s(const nefii_mlp &m, SStage<FT> (&b)[4], PCursor &cur) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int before, G;
    s_stream_geometry<FT>(m, before, G);
    const half8 *base = reinterpret_cast<const half8 *>(m.w_stream) + (size_t)8 * before * 256;
    cur.bytes = (unsigned)G * FT * 1024;
    cur.base = base + (size_t)wave * G * FT * 64 + lane;
    cur.off = 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) sload<FT>(b[u], cur);
}

// One tile of 16 * QT queries through the whole SDF network in a single fp16 pass - the BIG-tile form (96 / 128 queries,
// NEFII_COARSE_QT): one activation image, activation fragments read single-buffered and the whole epilogue behind the
// barrier to fit 256 registers.  The default 64- / 96-row tiles run sdf_tile16s2 below.
template <int QT, int FT, bool DB = false>
__device__ __forceinline__ void sdf_tile16s(const nefii_mlp &m, LdsS<FT, 16 * QT> &lds, float *raw, float *const *dest,
                                            SStage<FT> (&b)[4], PCursor &cur) {
    constexpr int NW = 8, RT = (QT + 1) / 2, XP = QGeo<FT>::XP, EP = QGeo<FT>::HW, RMAX = 16 * QT;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NH = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    const float k16 = inv_scale * A16_SCALE;
    const int boff = 16 * FT * wave + (lane & (16 * FT - 1));
    float bnext = m.layer[0].bias[boff];        // biases run one layer ahead (see sdf_tile16q)
    encode_tile16s<FT, 16 * QT>(m, raw, lds, 16 * QT);
    __syncthreads();
    const _Float16 *qh0 = lds.Xh + (lane & 15) * XP + 8 * (lane >> 4);
    for (int l = 0; l < NH; ++l) {
        const nefii_layer &L = m.layer[l];
        const int units = s_units(L);
        const _Float16 *ah = qh0 + (EP - L.k_x);
        const float *bp = m.layer[l + 1 < NH ? l + 1 : l].bias + boff;
        asm volatile("" ::"s"(units), "v"(ah), "v"(bp));
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): known state for the waitcnt pass (see 16p)
        __builtin_amdgcn_sched_barrier(0);
        const float bvec = bnext;
        f32x4 acc[FT * QT];
#pragma unroll
        for (int j = 0; j < FT * QT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
        static_assert(!DB, "the double-buffered tiles are sdf_tile16s2's");
        SAct<QT> a[1];
        NEFII_STAMP(0);
        sgemm<QT, FT, false>(units, b, a, cur, ah, acc);
        bnext = *bp;
        __builtin_amdgcn_sched_barrier(0);
        NEFII_STAMP(1);
        _Float16 *xh = lds.Xh + (EP - L.n_pad);
        // big tiles: no registers to park the packed results in - the whole epilogue runs behind the barrier, each
        // feature tile's values stored as they are produced
        NEFII_STAMP(2);
        __syncthreads();
        NEFII_STAMP(3);
        const int bsrc = __builtin_bit_cast(int, bvec * A16_SCALE);
        auto body = [&](auto fast) {        // the activation id resolved once per layer, not per value (see pepilogue)
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                const int f0 = 16 * FT * wave + 16 * ft + 4 * (lane >> 4);
                float4v bs;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    bs[k] = __builtin_bit_cast(float,
                                               __builtin_amdgcn_ds_bpermute(4 * (16 * ft + 4 * (lane >> 4) + k), bsrc));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const f32x4 &av = acc[ft * QT + qt];
                    half4 packed;
                    if constexpr (decltype(fast)::value) {
                        packed = softplus100_s16_pk4(av, k16, bs);
                    } else {
                        float4v hs;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float zs = __builtin_fmaf(av[k], k16, bs[k]);
                            hs[k] = act_fwd(zs * (1.f / A16_SCALE), m.act) * A16_SCALE;
                        }
                        packed = __builtin_convertvector(hs, half4);
                    }
                    const int query = 16 * qt + (lane & 15);
                    *reinterpret_cast<half4 *>(xh + query * XP + f0) = packed;
                }
            }
        };
        if (m.act == NEFII_ACT_SOFTPLUS100)
            body(std::true_type{});
        else
            body(std::false_type{});
        __syncthreads();
        NEFII_STAMP(4);
    }
    // last layer, column 0 only: hi fragments of the layer's own w_f16x3 (32x32x16), K split over the waves
    {
        const int r = lane & 31, h = lane >> 5;
        const nefii_layer &L = m.layer[NH];
        const int NT = L.n_pad >> 5;
        const half8 *wl = reinterpret_cast<const half8 *>(L.w_f16x3) + lane;
        const _Float16 *ah = lds.Xh + r * XP + 8 * h + (EP - L.k_x);
        const int ksw = (L.k_x >> 4) / NW;
        f32x16 acc2[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[rt][i] = 0.f;
        for (int u = 0; u < ksw; ++u) {
            const int s = wave * ksw + u;
            const half8 wh = wl[(size_t)s * NT * 128];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const half8 xh8 = *reinterpret_cast<const half8 *>(ah + rt * 32 * XP + 16 * s);
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh8, acc2[rt], 0, 0, 0);
            }
        }
        if (h == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                if (32 * rt + r < RMAX) raw[wave * RMAX + 32 * rt + r] = acc2[rt][0];
        }
        __syncthreads();
        if (threadIdx.x < 16 * QT) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += raw[w * RMAX + threadIdx.x];
            float *d = dest[threadIdx.x];
            if (d) *d = sum * inv_scale + L.bias[0];
        }
        __syncthreads();
    }
}

// "16s" with TWO activation images (the hi-only image of a 64- / 96-row tile is 75 KB: both fit the LDS): layer l reads
// image l & 1 and writes the other one, so the epilogue stores each packed value as it is produced and the barrier that
// made the stores wait for every wave to be done reading falls away - one barrier per layer instead of two, no parked
// results (32 registers less).  The encoding columns live in both images (the skip layer reads them from whichever
// image its hidden block is in).
template <int FT, int ROWS>
struct LdsS2 {
    _Float16 X[2][ROWS * QGeo<FT>::XP];
    _Float16 tail[128];
};

template <int QT, int FT>
__device__ __forceinline__ void sdf_tile16s2(const nefii_mlp &m, LdsS2<FT, 16 * QT> &lds, float *raw, float *const *dest,
                                             SStage<FT> (&b)[4], PCursor &cur) {
    constexpr int NW = 8, RT = (QT + 1) / 2, XP = QGeo<FT>::XP, EP = QGeo<FT>::HW, EW = QGeo<FT>::EW, RMAX = 16 * QT;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NH = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    const float k16 = inv_scale * A16_SCALE;
    const int boff = 16 * FT * wave + (lane & (16 * FT - 1));
    float bnext = m.layer[0].bias[boff];        // biases run one layer ahead (see sdf_tile16q)
    {
        const int w0 = enc_width(m.enc_freqs[0]);
        for (int i = threadIdx.x; i < RMAX * EW; i += 512) {
            const int p = i / EW, c = i - p * EW;
            const _Float16 v = (_Float16)((c < w0 ? enc_value(raw + p * 9, c) : 0.f) * A16_SCALE);
            lds.X[0][p * XP + EP + c] = v;
            lds.X[1][p * XP + EP + c] = v;
        }
    }
    __syncthreads();
    const int qoff = (lane & 15) * XP + 8 * (lane >> 4);
    for (int l = 0; l < NH; ++l) {
        const nefii_layer &L = m.layer[l];
        const int units = s_units(L);
        const _Float16 *ah = lds.X[l & 1] + qoff + (EP - L.k_x);
        _Float16 *xh = lds.X[(l & 1) ^ 1] + (EP - L.n_pad);
        const float *bp = m.layer[l + 1 < NH ? l + 1 : l].bias + boff;
        asm volatile("" ::"s"(units), "v"(ah), "v"(xh), "v"(bp));
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): known state for the waitcnt pass (see 16p)
        __builtin_amdgcn_sched_barrier(0);
        const float bvec = bnext;
        f32x4 acc[FT * QT];
#pragma unroll
        for (int j = 0; j < FT * QT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
        SAct<QT> a[2];
        NEFII_STAMP(0);
        sload_a<QT, XP>(a[0], ah, 0);
        sgemm<QT, FT, true>(units, b, a, cur, ah, acc);
        bnext = *bp;
        __builtin_amdgcn_sched_barrier(0);
        NEFII_STAMP(1);
        const int bsrc = __builtin_bit_cast(int, bvec * A16_SCALE);
        auto body = [&](auto fast) {        // the activation id resolved once per layer, not per value (see pepilogue)
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                const int f0 = 16 * FT * wave + 16 * ft + 4 * (lane >> 4);
                float4v bs;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    bs[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (16 * ft + 4 * (lane >> 4) + k), bsrc));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const f32x4 &av = acc[ft * QT + qt];
                    half4 packed;
                    if constexpr (decltype(fast)::value) {
                        packed = softplus100_s16_pk4(av, k16, bs);
                    } else {
                        float4v hs;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float zs = __builtin_fmaf(av[k], k16, bs[k]);
                            hs[k] = act_fwd(zs * (1.f / A16_SCALE), m.act) * A16_SCALE;
                        }
                        packed = __builtin_convertvector(hs, half4);
                    }
                    *reinterpret_cast<half4 *>(xh + (16 * qt + (lane & 15)) * XP + f0) = packed;
                }
            }
        };
        if (m.act == NEFII_ACT_SOFTPLUS100)
            body(std::true_type{});
        else
            body(std::false_type{});
        NEFII_STAMP(2);
        __syncthreads();
        NEFII_STAMP(3);
        NEFII_STAMP(4);
    }
    // last layer, column 0 only: hi fragments of the layer's own w_f16x3 (32x32x16), K split over the waves
    {
        const int r = lane & 31, h = lane >> 5;
        const nefii_layer &L = m.layer[NH];
        const int NT = L.n_pad >> 5;
        const half8 *wl = reinterpret_cast<const half8 *>(L.w_f16x3) + lane;
        const _Float16 *ah = lds.X[NH & 1] + r * XP + 8 * h + (EP - L.k_x);
        const int ksw = (L.k_x >> 4) / NW;
        f32x16 acc2[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[rt][i] = 0.f;
        for (int u = 0; u < ksw; ++u) {
            const int s = wave * ksw + u;
            const half8 wh = wl[(size_t)s * NT * 128];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const half8 xh8 = *reinterpret_cast<const half8 *>(ah + rt * 32 * XP + 16 * s);
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh8, acc2[rt], 0, 0, 0);
            }
        }
        if (h == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                if (32 * rt + r < RMAX) raw[wave * RMAX + 32 * rt + r] = acc2[rt][0];
        }
        __syncthreads();
        if (threadIdx.x < 16 * QT) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += raw[w * RMAX + threadIdx.x];
            float *d = dest[threadIdx.x];
            if (d) *d = sum * inv_scale + L.bias[0];
        }
        __syncthreads();
    }
}

// ================================================================================================
// "16d": the single-pass tile on FOUR waves - two independent four-wave GROUPS per workgroup (round 5).
// "16s" keeps all eight waves of a CU in lockstep: a layer is k-loop -> epilogue -> barrier for every wave, so the two
// waves of a SIMD want the matrix pipe together and the vector ALU together (pipe busy 54 %, vector issue 50 %,
// co-execution 17 %), and its two activation images fill the LDS.  Here a group is 4 waves (one per SIMD) over ONE
// activation image (75 KB), written in place behind a barrier; the two groups of a workgroup work on different tiles and
// nothing synchronises them - a group's barriers are its own (GroupBarrier: an LDS counter, not s_barrier) - so one group's
// epilogue and waits sit in the other's k-loop.  (Two 4-wave workgroups per CU do the same, but a CU that holds only one of
// them has room for foreign waves on its SIMDs: section 4b's register claim needs all eight waves in one workgroup.)
// A wave owns 128 features (the streams of "16s" waves 2w and 2w + 1, read alternately: unit u = stream u & 1, k-step
// u >> 1; same 4 KiB unit, same 4 register stages), so an activation fragment read feeds 8 MFMAs instead of 4: half the LDS
// reads per query.  The accumulation order of every output and the last layer's eight K partitions are those of "16s": the
// values are bit-identical.
// ================================================================================================
// barrier of one four-wave group: every wave adds 1 to the group's LDS counter and waits for 4 x (barriers so far).  LDS
// operations of a wave complete in order, so the stores before the add are visible to whoever sees the add.
typedef __attribute__((address_space(3))) unsigned lds_u32;
struct GroupBarrier {
    lds_u32 *ctr;
    unsigned phase;
    __device__ __forceinline__ GroupBarrier(unsigned *shared_counter) : ctr((lds_u32 *)shared_counter), phase(0u) {}
    __device__ __forceinline__ void sync() {
        __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): this wave's LDS stores are done (the prefetches stay in flight)
        asm volatile("" ::: "memory");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        phase += 4;
        while ((int)(__builtin_amdgcn_readfirstlane(*(volatile lds_u32 *)ctr) - phase) < 0) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    }
    // end of a group's work: its waves stay resident (and keep their registers: section 4b's claim) until all eight are done
    static __device__ __forceinline__ void hold(unsigned *shared_done) {
        lds_u32 *d = (lds_u32 *)shared_done;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(d, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__builtin_amdgcn_readfirstlane(*(volatile lds_u32 *)d) < 8u) __builtin_amdgcn_s_sleep(8);
    }
};

template <int QT, int J>
__device__ __forceinline__ void dstep(SStage<4> (&b)[4], SAct<QT> (&a)[2], PCursor (&cur)[2], const _Float16 *ah, int s32,
                                      f32x4 (&acc)[8 * QT]) {
    constexpr int HALF = J & 1, AB = (J >> 1) & 1;
    sload<4>(b[(J + 3) % 4], cur[(J + 3) & 1]);
    if constexpr (HALF == 0) sload_a<QT, QGeo<4>::XP>(a[AB ^ 1], ah, s32 + 1);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 &c = acc[(4 * HALF + ft) * QT + qt];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[J].f[ft], a[AB].h[qt], c, 0, 0, 0);
        }
#define NEFII_DGROUP(i)                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, QT / 2, 0);                                       \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, QT - QT / 2, 0);                                  \
    if constexpr (HALF == 0) __builtin_amdgcn_sched_group_barrier(0x100, (QT * ((i) + 1)) / 4 - (QT * (i)) / 4, 0);
    NEFII_DGROUP(0)
    NEFII_DGROUP(1)
    NEFII_DGROUP(2)
    NEFII_DGROUP(3)
#undef NEFII_DGROUP
    __builtin_amdgcn_sched_barrier(0);
}

template <int QT>
__device__ __forceinline__ void prime16d(const nefii_mlp &m, SStage<4> (&b)[4], PCursor (&cur)[2]) {
    const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3), lane = threadIdx.x & 63;
    int before, G;
    s_stream_geometry<4>(m, before, G);
    const half8 *base = reinterpret_cast<const half8 *>(m.w_stream) + (size_t)8 * before * 256;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        cur[h].bytes = (unsigned)G * 4 * 1024;
        cur[h].base = base + (size_t)(2 * wave + h) * G * 4 * 64 + lane;
        cur[h].off = 0;
    }
    sload<4>(b[0], cur[0]);
    sload<4>(b[1], cur[1]);
    sload<4>(b[2], cur[0]);
}

#ifdef NEFII_STAMPS     /* wave 0 of every workgroup, its third tile: 5 stamps per layer + the hardware id of its CU */
__device__ unsigned long long g_dstamps[512 * 12 * 5];
__device__ unsigned g_dhwid[512 * 2];
#define NEFII_DSTAMP(i)                                                                                  \
    if (tile_seq == 2 && tl == 0 && blockIdx.x < 256)                                                     \
        g_dstamps[((2 * blockIdx.x + (threadIdx.x >> 8)) * 12 + l) * 5 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define NEFII_DSTAMP(i)
#endif
template <int QT>
__device__ __forceinline__ void sdf_tile16d(const nefii_mlp &m, _Float16 *X, float *raw, float *const *dest,
                                            SStage<4> (&b)[4], PCursor (&cur)[2], GroupBarrier &gb, int tile_seq = 0) {
    // X, raw, dest: this group's activation image, decoded queries and destinations
    constexpr int RT = (QT + 1) / 2, XP = QGeo<4>::XP, EP = QGeo<4>::HW, EW = QGeo<4>::EW, RMAX = 16 * QT;
    const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3), lane = threadIdx.x & 63;
    const int tl = threadIdx.x & 255;
    const int NH = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    const float k16 = inv_scale * A16_SCALE;
    const int boff = 128 * wave + lane;
    float bnext0 = m.layer[0].bias[boff], bnext1 = m.layer[0].bias[boff + 64];   // biases run one layer ahead
    {
        const int w0 = enc_width(m.enc_freqs[0]);
        for (int i = tl; i < RMAX * EW; i += 256) {
            const int p = i / EW, c = i - p * EW;
            X[p * XP + EP + c] = (_Float16)((c < w0 ? enc_value(raw + p * 9, c) : 0.f) * A16_SCALE);
        }
    }
    gb.sync();
    const int qoff = (lane & 15) * XP + 8 * (lane >> 4);
    for (int l = 0; l < NH; ++l) {
        const nefii_layer &L = m.layer[l];
        const int units = s_units(L);
        const _Float16 *ah = X + qoff + (EP - L.k_x);
        _Float16 *xh = X + (EP - L.n_pad);
        const float *bp = m.layer[l + 1 < NH ? l + 1 : l].bias + boff;
        asm volatile("" ::"s"(units), "v"(ah), "v"(xh), "v"(bp));
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): known state for the waitcnt pass (see 16p)
        __builtin_amdgcn_sched_barrier(0);
        const float bvec0 = bnext0, bvec1 = bnext1;
        f32x4 acc[8 * QT];
#pragma unroll
        for (int j = 0; j < 8 * QT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
        SAct<QT> a[2];
        NEFII_DSTAMP(0);
        sload_a<QT, XP>(a[0], ah, 0);
        for (int s = 0; s < units; s += 2) {
            dstep<QT, 0>(b, a, cur, ah, s, acc);
            dstep<QT, 1>(b, a, cur, ah, s, acc);
            dstep<QT, 2>(b, a, cur, ah, s + 1, acc);
            dstep<QT, 3>(b, a, cur, ah, s + 1, acc);
        }
        bnext0 = bp[0];
        bnext1 = bp[64];
        __builtin_amdgcn_sched_barrier(0);
        NEFII_DSTAMP(1);
        gb.sync();        // every wave is done reading the image: the epilogue writes it in place
        NEFII_DSTAMP(2);
        auto body = [&](auto fast) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int bsrc = __builtin_bit_cast(int, (h ? bvec1 : bvec0) * A16_SCALE);
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    const int f0 = 128 * wave + 64 * h + 16 * ft + 4 * (lane >> 4);
                    float4v bs;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        bs[k] = __builtin_bit_cast(float,
                                                   __builtin_amdgcn_ds_bpermute(4 * (16 * ft + 4 * (lane >> 4) + k), bsrc));
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const f32x4 &av = acc[(4 * h + ft) * QT + qt];
                        half4 packed;
                        if constexpr (decltype(fast)::value) {
                            packed = softplus100_s16_pk4(av, k16, bs);
                        } else {
                            float4v hs;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float zs = __builtin_fmaf(av[k], k16, bs[k]);
                                hs[k] = act_fwd(zs * (1.f / A16_SCALE), m.act) * A16_SCALE;
                            }
                            packed = __builtin_convertvector(hs, half4);
                        }
                        *reinterpret_cast<half4 *>(xh + (16 * qt + (lane & 15)) * XP + f0) = packed;
                    }
                }
            }
        };
        if (m.act == NEFII_ACT_SOFTPLUS100)
            body(std::true_type{});
        else
            body(std::false_type{});
        NEFII_DSTAMP(3);
        gb.sync();
        NEFII_DSTAMP(4);
    }
    // last layer, column 0 only: the eight K partitions of "16s", two per wave, summed in its order
    {
        const int r = lane & 31, h = lane >> 5;
        const nefii_layer &L = m.layer[NH];
        const int NT = L.n_pad >> 5;
        const half8 *wl = reinterpret_cast<const half8 *>(L.w_f16x3) + lane;
        const _Float16 *ah = X + r * XP + 8 * h + (EP - L.k_x);
        const int ksw = (L.k_x >> 4) / 8;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int part = 2 * wave + p;
            f32x16 acc2[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[rt][i] = 0.f;
            for (int u = 0; u < ksw; ++u) {
                const int s = part * ksw + u;
                const half8 wh = wl[(size_t)s * NT * 128];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const half8 xh8 = *reinterpret_cast<const half8 *>(ah + rt * 32 * XP + 16 * s);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh8, acc2[rt], 0, 0, 0);
                }
            }
            if (h == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    if (32 * rt + r < RMAX) raw[part * RMAX + 32 * rt + r] = acc2[rt][0];
            }
        }
        gb.sync();
        if (tl < 16 * QT) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += raw[w * RMAX + tl];
            float *d = dest[tl];
            if (d) *d = sum * inv_scale + L.bias[0];
        }
        gb.sync();
    }
}


// ================================================================================================
// "16f" (round 6): the split-precision evaluator with its two CORRECTION products on block-scaled fp8.
//   x w ~ x_h w_h  +  q(x_h) q(w_l)  +  q(x_l) q(w_h),      q = OCP e4m3 with ONE constant power-of-two scale per operand kind
// The main product stays on v_mfma_f32_16x16x32_f16; each correction product of a 128-deep chunk is ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 (4 x the K of the fp16 form in 2 x its cycles): 4 + 2 x 2 = 8 fp16-MFMA times per chunk and
// (feature tile, query tile) instead of 12, and the fp8 form costs the power-limited part less energy per flop
// (profiles/r06/fp8_probe.txt: a bare chunk loop 14.6 -> 10.0 ms).  The correction terms are 2^-11 of the product, so an e4m3
// operand (2^-4 relative) leaves 2^-15: max |sdf error| against fp64 1.0e-5 (1.6e-6 within 0.02 of the surface; "16q": 5e-7;
// tools/experiments/arith_emulation.py `fp8corr_fix`).  No data-dependent block scales: a low part is bounded by the ulp of its
// high part and what falls below e4m3's range is negligible in absolute terms - the constants below, restored by the
// instruction's scale operand.  512-wide nets only (FT = 4); the last layer's single column is the fp16 split of "16q": behind the
// last hidden layer the epilogue writes an fp16 lo image over the two fp8 images, which nothing reads any more in that tile.
// Stream (nefii_pack_sdf_stream's fifth copy): per wave and 128-deep chunk of a layer's 128-padded K eight 4-KiB units - the
// chunk's four 32-deep k-steps in hi fragments (as "16s"), then per feature tile [q(w_l) 32 B | q(w_h) 32 B] per lane, lane
// (f = lane & 15, g = lane >> 4) holding k = 128 chunk + 32 g + 0..31 - one cursor, the same 4 register stages.
// LDS: the fp16 hi image of "16q", and instead of its lo image two fp8 images (q(x_l), q(x_h)) of XP8 bytes per row.
// ================================================================================================
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int F8_XH_E = -3, F8_WL_E = 10, F8_XL_E = 8, F8_WH_E = -1;       // stored value = operand x 2^E (operands carry x 16 / x 64)
constexpr int F8_SCALE_HL = (127 - (F8_XH_E + F8_WL_E)) * 0x01010101;        // E8M0 byte (in every byte lane of the scale operand) that restores q(x_h) q(w_l): 2^-7
constexpr int F8_SCALE_LH = (127 - (F8_XL_E + F8_WH_E)) * 0x01010101;        // ... and q(x_l) q(w_h): 2^-7
constexpr int F8_SCALE_ONE = 0x7f7f7f7f;
constexpr int XP8 = 592;                                                   // bytes per row of an fp8 image (16 B * 37)
struct LdsF {
    _Float16 Xh[64 * QGeo<4>::XP];
    // order matters: a layer's 128-padded K reads past the end of a row - past the LAST row into what follows - and multiplies it
    // by zero weights: q(x_l) bytes read as halves are finite (|x_l| 2^8 < 2^7: exponent field never all ones), any e4m3 byte the
    // epilogue writes is finite, the tail is zero
    unsigned char Fl[64 * XP8], Fh[64 * XP8];
    _Float16 tail[128];
};
__device__ __forceinline__ int f8_pack4(float a, float b, float c, float d) {
    const float lim = 448.f;
    a = __builtin_fminf(__builtin_fmaxf(a, -lim), lim), b = __builtin_fminf(__builtin_fmaxf(b, -lim), lim);
    c = __builtin_fminf(__builtin_fmaxf(c, -lim), lim), d = __builtin_fminf(__builtin_fmaxf(d, -lim), lim);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}
__device__ __forceinline__ float f8_scale(int e) { return __builtin_bit_cast(float, (127 + e) << 23); }
// chunks (128-deep) of a layer in the fifth stream copy; 8 units each
__host__ __device__ __forceinline__ int f_chunks(const nefii_layer &L) { return (L.k_x + L.k_e + 127) >> 7; }

template <int QT>
struct FAct {
    i32x8 h[QT], l[QT];
};
template <int QT>
__device__ __forceinline__ void fload_x8(FAct<QT> &st, const unsigned char *fh, const unsigned char *fl, int chunk) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const i32x4 *ph = reinterpret_cast<const i32x4 *>(fh + qt * 16 * XP8 + 128 * chunk);
        const i32x4 *pl = reinterpret_cast<const i32x4 *>(fl + qt * 16 * XP8 + 128 * chunk);
        const i32x4 h0 = ph[0], h1 = ph[1], l0 = pl[0], l1 = pl[1];
        st.h[qt] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        st.l[qt] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// main step J (0..3) of a chunk: the hi x hi MFMAs of one 32-deep k-step, as "16s"; LOADX: also fetch the chunk's fp8 activations
template <int QT, int J, bool LOADX>
__device__ __forceinline__ void fstep_main(SStage<4> (&b)[4], SAct<QT> (&a)[2], FAct<QT> &x8, PCursor &cur, const _Float16 *ah,
                                           const unsigned char *fh, const unsigned char *fl, int chunk, f32x4 (&acc)[4 * QT]) {
    sload<4>(b[(J + 3) % 4], cur);
    sload_a<QT, QGeo<4>::XP>(a[(J + 1) & 1], ah, 4 * chunk + J + 1);
    if (LOADX) fload_x8<QT>(x8, fh, fl, chunk);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 &c = acc[ft * QT + qt];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[J].f[ft], a[J & 1].h[qt], c, 0, 0, 0);
        }
    // MFMAs lead; the unit's four fragment loads and the LDS reads (QT of the next k-step, with LOADX 4 QT more) are spread between them
#define NEFII_FGROUP(i)                                                                                        \
    __builtin_amdgcn_sched_group_barrier(0x008, QT / 2, 0);                                                    \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, QT - QT / 2, 0);                                               \
    __builtin_amdgcn_sched_group_barrier(0x100, (LOADX ? QT : 0) + (QT * ((i) + 1)) / 4 - (QT * (i)) / 4, 0);
    NEFII_FGROUP(0)
    NEFII_FGROUP(1)
    NEFII_FGROUP(2)
    NEFII_FGROUP(3)
#undef NEFII_FGROUP
    __builtin_amdgcn_sched_barrier(0);
}
// correction step of feature tile FTI: stage (4 + FTI) % 4 = FTI holds [q(w_l) | q(w_h)] of that tile
template <int QT, int FTI>
__device__ __forceinline__ void fstep_corr(SStage<4> (&b)[4], const FAct<QT> &x8, PCursor &cur, f32x4 (&acc)[4 * QT]) {
    sload<4>(b[(FTI + 3) % 4], cur);
    const i32x4 w0 = __builtin_bit_cast(i32x4, b[FTI].f[0]), w1 = __builtin_bit_cast(i32x4, b[FTI].f[1]);
    const i32x4 w2 = __builtin_bit_cast(i32x4, b[FTI].f[2]), w3 = __builtin_bit_cast(i32x4, b[FTI].f[3]);
    const i32x8 wl8 = __builtin_shufflevector(w0, w1, 0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8 wh8 = __builtin_shufflevector(w2, w3, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        f32x4 &c = acc[FTI * QT + qt];
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wl8, x8.h[qt], c, 0, 0, 0, F8_SCALE_HL, 0, F8_SCALE_ONE);
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wh8, x8.l[qt], c, 0, 0, 0, F8_SCALE_LH, 0, F8_SCALE_ONE);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, QT, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, QT, 0);
    __builtin_amdgcn_sched_barrier(0);
}
template <int QT>
__device__ __forceinline__ void fgemm(int chunks, SStage<4> (&b)[4], SAct<QT> (&a)[2], FAct<QT> &x8, PCursor &cur,
                                      const _Float16 *ah, const unsigned char *fh, const unsigned char *fl, f32x4 (&acc)[4 * QT]) {
    for (int c = 0; c < chunks; ++c) {
        fstep_main<QT, 0, true>(b, a, x8, cur, ah, fh, fl, c, acc);
        fstep_main<QT, 1, false>(b, a, x8, cur, ah, fh, fl, c, acc);
        fstep_main<QT, 2, false>(b, a, x8, cur, ah, fh, fl, c, acc);
        fstep_main<QT, 3, false>(b, a, x8, cur, ah, fh, fl, c, acc);
        fstep_corr<QT, 0>(b, x8, cur, acc);
        fstep_corr<QT, 1>(b, x8, cur, acc);
        fstep_corr<QT, 2>(b, x8, cur, acc);
        fstep_corr<QT, 3>(b, x8, cur, acc);
    }
}

template <int QT, bool FAST>
__device__ __forceinline__ void fepilogue(const f32x4 (&acc)[4 * QT], float bvec, float k16, int lane, int act,
                                          half4 (&phi)[4 * QT], int (&p8h)[4 * QT], int (&p8l)[4 * QT]) {
    const int bsrc = __builtin_bit_cast(int, bvec * A16_SCALE);
    const float sxh = f8_scale(F8_XH_E), sxl = f8_scale(F8_XL_E);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        float4v bs;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            bs[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (16 * ft + 4 * (lane >> 4) + k), bsrc));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const f32x4 &av = acc[ft * QT + qt];
            float4v hs;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float zs = __builtin_fmaf(av[k], k16, bs[k]);
                hs[k] = FAST ? softplus100_s16(zs) : act_fwd(zs * (1.f / A16_SCALE), act) * A16_SCALE;
            }
            const half4 hi = __builtin_convertvector(hs, half4);
            const float4v hf = __builtin_convertvector(hi, float4v), lf = hs - hf;
            phi[ft * QT + qt] = hi;
            p8h[ft * QT + qt] = f8_pack4(hf[0] * sxh, hf[1] * sxh, hf[2] * sxh, hf[3] * sxh);
            p8l[ft * QT + qt] = f8_pack4(lf[0] * sxl, lf[1] * sxl, lf[2] * sxl, lf[3] * sxl);
        }
    }
}

// stages 0..2 <- units 0..2 of the wave's stream (start of a workgroup); f8_stream: the fifth copy (host: f8_stream_offset)
__device__ __forceinline__ void prime16f(const nefii_mlp &m, const void *f8_stream, SStage<4> (&b)[4], PCursor &cur) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int G = 0;
    for (int l = 0; l < m.n_layers - 1; ++l) G += 8 * f_chunks(m.layer[l]);
    cur.bytes = (unsigned)G * 4096;
    cur.base = reinterpret_cast<const half8 *>(f8_stream) + (size_t)wave * G * 256 + lane;
    cur.off = 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) sload<4>(b[u], cur);
}

// One tile of 16 * QT queries (QT = 4 / 2) through the whole 512-wide SDF network; interface of sdf_tile16q
template <int QT>
__device__ __forceinline__ void sdf_tile16f(const nefii_mlp &m, LdsF &lds, float *raw, float *const *dest, SStage<4> (&b)[4],
                                            PCursor &cur, const float *coarse_old = nullptr, int *audit = nullptr) {
    constexpr int NW = 8, FT = 4, RT = QT / 2, XP = QGeo<4>::XP, EP = QGeo<4>::HW, EW = QGeo<4>::EW, RMAX = QGeo<4>::ROWS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NH = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    const float k16 = inv_scale * A16_SCALE;
    const float sxh = f8_scale(F8_XH_E), sxl = f8_scale(F8_XL_E);
    const int boff = 16 * FT * wave + (lane & (16 * FT - 1));
    float bnext = m.layer[0].bias[boff];
    {       // positional encoding of the queries: hi halves and the fp8 images of hi and lo
        const int w0 = enc_width(m.enc_freqs[0]);
        for (int i = threadIdx.x; i < 16 * QT * EW; i += 512) {
            const int p = i / EW, c = i - p * EW;
            const float val = c < w0 ? enc_value(raw + p * 9, c) : 0.f;
            _Float16 hi, lo;
            split16a(val, hi, lo);
            lds.Xh[p * XP + EP + c] = hi;
            lds.Fh[p * XP8 + EP + c] = (unsigned char)(f8_pack4((float)hi * sxh, 0.f, 0.f, 0.f) & 0xff);
            lds.Fl[p * XP8 + EP + c] = (unsigned char)(f8_pack4((float)lo * sxl, 0.f, 0.f, 0.f) & 0xff);
        }
    }
    __syncthreads();
    const int qoff = (lane & 15) * XP + 8 * (lane >> 4), qoff8 = (lane & 15) * XP8 + 32 * (lane >> 4);
    for (int l = 0; l < NH; ++l) {
        const nefii_layer &L = m.layer[l];
        const int chunks = f_chunks(L);
        const _Float16 *ah = lds.Xh + qoff + (EP - L.k_x);
        const unsigned char *fh = lds.Fh + qoff8 + (EP - L.k_x), *fl = lds.Fl + qoff8 + (EP - L.k_x);
        const float *bp = m.layer[l + 1 < NH ? l + 1 : l].bias + boff;
        asm volatile("" ::"s"(chunks), "v"(ah), "v"(fh), "v"(fl), "v"(bp));
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): known state for the waitcnt pass (see 16p)
        __builtin_amdgcn_sched_barrier(0);
        const float bvec = bnext;
        f32x4 acc[FT * QT];
#pragma unroll
        for (int j = 0; j < FT * QT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
        SAct<QT> a[2];
        FAct<QT> x8;
        sload_a<QT, XP>(a[0], ah, 0);
        fgemm<QT>(chunks, b, a, x8, cur, ah, fh, fl, acc);
        bnext = *bp;
        __builtin_amdgcn_sched_barrier(0);
        // epilogue: bias, activation, split into the fp16 hi half and the fp8 images of hi and lo - or, behind the LAST hidden layer,
        // into fp16 hi and lo: the last layer's single column is computed in the fp16 split (below), and the two fp8 images, which
        // nothing reads any more in this tile, hold its lo image meanwhile (same 75 KB)
        if (l < NH - 1) {
            half4 phi[FT * QT];
            int p8h[FT * QT], p8l[FT * QT];
            if (m.act == NEFII_ACT_SOFTPLUS100)
                fepilogue<QT, true>(acc, bvec, k16, lane, m.act, phi, p8h, p8l);
            else
                fepilogue<QT, false>(acc, bvec, k16, lane, m.act, phi, p8h, p8l);
            __syncthreads();
            _Float16 *xh = lds.Xh + (EP - L.n_pad);
            unsigned char *x8h = lds.Fh + (EP - L.n_pad), *x8l = lds.Fl + (EP - L.n_pad);
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                const int f0 = 16 * FT * wave + 16 * ft + 4 * (lane >> 4);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int query = 16 * qt + (lane & 15);
                    *reinterpret_cast<half4 *>(xh + query * XP + f0) = phi[ft * QT + qt];
                    *reinterpret_cast<int *>(x8h + query * XP8 + f0) = p8h[ft * QT + qt];
                    *reinterpret_cast<int *>(x8l + query * XP8 + f0) = p8l[ft * QT + qt];
                }
            }
        } else {
            half4 phi[FT * QT], plo[FT * QT];
            if (m.act == NEFII_ACT_SOFTPLUS100)
                qepilogue<QT, FT, true>(acc, bvec, k16, lane, m.act, phi, plo);
            else
                qepilogue<QT, FT, false>(acc, bvec, k16, lane, m.act, phi, plo);
            __syncthreads();
            _Float16 *xh = lds.Xh + (EP - L.n_pad), *xl = reinterpret_cast<_Float16 *>(lds.Fl) + (EP - L.n_pad);
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                const int f0 = 16 * FT * wave + 16 * ft + 4 * (lane >> 4);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int query = 16 * qt + (lane & 15);
                    *reinterpret_cast<half4 *>(xh + query * XP + f0) = phi[ft * QT + qt];
                    *reinterpret_cast<half4 *>(xl + query * XP + f0) = plo[ft * QT + qt];
                }
            }
        }
        __syncthreads();
    }
    // last layer, column 0 only: the fp16 split of "16q" (w_h x_h + w_l x_h + w_h x_l on 32x32x16 fragments of the layer's own w_f16x3,
    // K split over the waves) - the x_l image sits where the fp8 images were.  (A geometric-init net has an all-positive last layer:
    // an e4m3 x_l there gave errors that do not cancel - 5.8e-6 near the surface instead of 1.5e-6.)
    {
        static_assert(sizeof(lds.Fl) + sizeof(lds.Fh) >= sizeof(_Float16) * 64 * QGeo<4>::XP, "the lo image must fit the two fp8 images");
        const int r = lane & 31, h = lane >> 5;
        const nefii_layer &L = m.layer[NH];
        const int NT = L.n_pad >> 5;
        const half8 *wl = reinterpret_cast<const half8 *>(L.w_f16x3) + lane;
        const _Float16 *ah = lds.Xh + r * XP + 8 * h + (EP - L.k_x);
        const _Float16 *al = reinterpret_cast<const _Float16 *>(lds.Fl) + r * XP + 8 * h + (EP - L.k_x);
        const int ksw = (L.k_x >> 4) / NW;
        f32x16 acc2[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[rt][i] = 0.f;
        for (int u = 0; u < ksw; ++u) {
            const int s = wave * ksw + u;
            const half8 wh = wl[(size_t)s * NT * 128], wlo = wl[(size_t)s * NT * 128 + 64];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const half8 xh8 = *reinterpret_cast<const half8 *>(ah + rt * 32 * XP + 16 * s);
                const half8 xl8 = *reinterpret_cast<const half8 *>(al + rt * 32 * XP + 16 * s);
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh8, acc2[rt], 0, 0, 0);
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xh8, acc2[rt], 0, 0, 0);
                acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl8, acc2[rt], 0, 0, 0);
            }
        }
        if (h == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) raw[wave * RMAX + 32 * rt + r] = acc2[rt][0];
        }
        __syncthreads();
        // the lo image has overwritten bytes that the NEXT tile's 128-padded first layer multiplies by zero weights before anything
        // rewrites them: the pad columns [576, 592) of every fp8 row and the 48 bytes behind them (the next row's first columns; past
        // the last row of Fl that is the head of Fh, past Fh the tail) - fp16 bits read as e4m3 may be NaN: zero them
        if (threadIdx.x < 128) {
            unsigned char *img = (threadIdx.x >> 6) ? lds.Fh : lds.Fl;
            i32x4 *z = reinterpret_cast<i32x4 *>(img + (threadIdx.x & 63) * XP8 + 576);
            const i32x4 zero = {0, 0, 0, 0};
            z[0] = zero, z[1] = zero, z[2] = zero, z[3] = zero;
        }
        float dm = 0.f;
        if (threadIdx.x < 32 * RT) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += raw[w * RMAX + threadIdx.x];
            float *d = dest[threadIdx.x];
            const float v = sum * inv_scale + L.bias[0];
            if (d) *d = v;
            if (coarse_old) {
                const float o = coarse_old[threadIdx.x];
                if (d && o == o) dm = __builtin_fabsf(o - v);
            }
        }
        if (coarse_old && threadIdx.x < 128) {
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) dm = __builtin_fmaxf(dm, __shfl_xor(dm, s));
            if ((threadIdx.x & 63) == 0 && dm > 0.f) atomicMax(audit, __builtin_bit_cast(int, dm));
        }
        __syncthreads();
    }
}

// accumulator element -> (row, col) of the 32 x n_pad output (32x32 C/D map: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); cdna_hip_programming.md section 3)
#define NEFII_FOR_ACC(acc, ntw, BODY)                                                     \
    {                                                                                     \
        const int _wave = threadIdx.x >> 6, _lane = threadIdx.x & 63;                     \
        _Pragma("unroll") for (int _j = 0; _j < 4; ++_j) if (_j < (ntw)) {                \
            const int col = 32 * (_wave + 4 * _j) + (_lane & 31);                         \
            _Pragma("unroll") for (int _i = 0; _i < 16; ++_i) {                           \
                const int row = (_i & 3) + 8 * (_i >> 2) + 4 * (_lane >> 5);              \
                const float val = (acc)[_j][_i];                                               \
                BODY                                                                      \
            }                                                                             \
        }                                                                                 \
    }

// nefii_sdf_value_grad on the pipelined fragment stream (defined beside the stream code in nefii_tracer.hip):
// value_grad_stream_ws_bytes: workspace of the streamed kernel for n points, 0 when the net does not take it;
// value_grad_stream_launch: NEFII_E_UNSUPPORTED when it does not.
size_t value_grad_stream_ws_bytes(const nefii_mlp *m, int64_t n);
int value_grad_stream_launch(const nefii_mlp *m, const float *x, int64_t n, float *sdf_out, int out_stride, float *feat_out,
                             int feat_stride, float *grad_out, float *ws, hipStream_t st);

}  // namespace nefii
