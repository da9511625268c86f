// sdf_tile_c.h - "16c": the tracer's single-pass (coarse) SDF evaluator, round 4's structure:
//     ACTIVATIONS STAY IN REGISTERS, WEIGHT FRAGMENTS ARE SHARED THROUGH AN LDS RING.
//
// Why (profiles/r04/pmc_classes_eval_tile.txt): in the "16s" tile (mlp_tile.h) a layer is k-loop -> epilogue -> barrier for
// all 8 waves together (the activations of a layer are exchanged through LDS), so the two waves of a SIMD want the matrix
// pipe at the same time and the vector ALU at the same time: SQ_VALU_MFMA_BUSY 54 % of the cycles, vector issue 50 %, and
// only 17 % of the matrix-busy cycles have a vector instruction executing beside them.  The dependency that forces the
// barrier is the layout, not the algorithm:
//   * with the WEIGHT fragment as the MFMA's A operand the accumulator of v_mfma_f32_32x32x16_f16 holds, per lane, 16
//     features of ONE query (column = lane & 31) - and the B operand of the next layer's MFMA wants, per lane, 8 K-values of
//     that same query.  With the output features of a tile stored in the order the next layer consumes them (a bit swap of
//     the row index, done once when the stream is packed) a finished 32-feature tile IS two 16-deep k-steps of the next
//     layer's B operand after the activation and one v_cvt_pk per pair: activations never leave the wave's registers;
//   * a wave then owns 32 queries through the whole network: no exchange, no per-layer barrier, and the epilogue of output
//     tile T (16 values per lane) is interleaved with the 32 MFMAs of tile T + 1 of the SAME wave - 8 of the 32 cycles of
//     each MFMA are issue, 24 are free for vector work (MI355X_MICROARCH.md, "vector-instruction ISSUE cost");
//   * what the waves share instead is the weight stream: every wave multiplies the same fragments.  They come from L2 ONCE
//     per workgroup - 4 waves x 32 queries = 128 queries per 3.6 MB instead of 64 per 3.9 MB - by LDS-DMA
//     (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction) into a 128-KiB ring of 1-KiB fragments, 24 KiB per wave
//     ahead of the reads, one s_barrier per 32 fragments (= per 32 MFMAs of every wave) and counted vmcnt waits; each
//     wave reads every fragment back with one ds_read_b128 per MFMA (128 B/clk/CU, half the LDS rate).
// One wave per SIMD (the register-resident input of a 512-wide layer is 128 registers, the output under construction
// another 128), 4 waves per workgroup, one workgroup per CU.
//
// Stream ("c" copy of nefii_mlp.w_stream, nefii_pack_sdf_stream): fragments of 1 KiB in consumption order -
//   layer 0:            tile T = 0 .. HW/32-1, k-step s = 0 .. 3 of the 64 encoding columns (39 used);
//   layers 1 .. NH-1:   tile T, k-step s = 0 .. HW/16-1 of the layer's HW inputs; the skip layer's input is the TRUE
//                       concatenation [hidden (HW - 39) | encoding (39)] (the reference's torch.cat order,
//                       implicit_differentiable_renderer.py:93-97), not the padded [HW | 64] image of the other kernels;
//   last layer:         one tile (row 0 = the SDF column), HW/16 k-steps;
// fragment (T, s): lane L holds hi16(W[F(T, L & 31)][16 s + 8 (L >> 5) + 0..7]) UNSCALED (the hi fragments of w_f16x3 carry
// x 64 for their lo halves' sake; here it would only cost a multiply per value), F(T, m) = 32 T + (m with bits 2 and 3
// swapped); then per hidden layer and tile the 2 x 16 bias values (x 16) in accumulator order.
// Arithmetic: one fp16 pass, fp32 accumulate, activations carried x 16 (A16_SCALE) as in "16s"; the accumulator starts at
// 16 x bias, so the epilogue is v_cvt_pk + the packed-fp16 softplus of mlp_tile.h (softplus100_s16_pk).  What this evaluator
// may get wrong is measured per network like "16s" (nefii_tracer_params.coarse_tau: ops.calibrate_coarse_tau runs whichever
// coarse evaluator the tracer will use).
#pragma once
#include "mlp_tile.h"

namespace nefii {

constexpr int C_WIN_FRAGS = 32;                 // fragments per ring window = per s_barrier
constexpr int C_RING_BYTES = 128 * 1024;        // 4 windows
constexpr int C_WIN_BYTES = C_WIN_FRAGS * 1024;
constexpr int C_AHEAD_WINS = 3;                 // the DMA fills window c + 3 while window c is read
constexpr int C_ROWS = 128;                     // queries per workgroup pass (4 waves x 32)
constexpr int C_MAX_HIDDEN = 12;

// geometry of the "c" copy for a net of the pipelined shapes (shape16p): fragments per pass, bias table
struct CStream {
    int hw;             // hidden width (512)
    int nh;             // hidden layers
    int skip;           // index of the skip layer (k_x > 0 and k_e > 0), -1: none
    int frags;          // fragments per pass (a multiple of C_WIN_FRAGS)
    int bias_bytes;     // nh * (hw / 32) * 128
};
__host__ __device__ __forceinline__ CStream c_stream_geometry(const nefii_mlp &m) {
    CStream g;
    g.hw = m.layer[0].n_pad;
    g.nh = m.n_layers - 1;
    g.skip = -1;
    const int nt = g.hw / 32, ks = g.hw / 16;
    int f = nt * 4;
    for (int l = 1; l < g.nh; ++l) {
        f += nt * ks;
        if (m.layer[l].k_e > 0) g.skip = l;
    }
    f += ks;
    g.frags = (f + C_WIN_FRAGS - 1) / C_WIN_FRAGS * C_WIN_FRAGS;
    g.bias_bytes = g.nh * nt * 128;
    return g;
}
__host__ __device__ __forceinline__ size_t c_stream_bytes(const nefii_mlp &m) {
    const CStream g = c_stream_geometry(m);
    return (size_t)g.frags * 1024 + g.bias_bytes;
}
// row m of a tile holds output feature 32 T + cperm(m): accumulator register 4 j + i of lane (n, h) is row 8 j + 4 h + i, and
// the next layer's k-step 2 T + u wants K = 16 u + 8 h + 4 (j & 1) + i there (u = j >> 1)
__host__ __device__ __forceinline__ int cperm(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

struct LdsC {
    char ring[C_RING_BYTES];
    float bias[C_MAX_HIDDEN * 16 * 32];         // [layer][tile][h][16]: 16 x bias in accumulator order (24 KiB at most)
};

// the workgroup's position in the stream; everything wave-uniform (SGPRs)
struct CRing {
    __amdgpu_buffer_rsrc_t srd;
    unsigned voff;          // lane * 16
    unsigned lds0;          // LDS byte address of the ring
    unsigned rd;            // ring offset of the window being read
    unsigned wr;            // ring offset of the window being filled
    unsigned src;           // stream offset of the window being filled
    unsigned pass_bytes;
    unsigned wave_off;      // wave * 1024: this wave's fragment of each group of 4
};

// one DMA piece: fragment 4 J + wave of the window being filled
template <int J>
__device__ __forceinline__ void c_dma(const CRing &r, unsigned dwin, unsigned swin) {
    unsigned tmp;
    asm volatile("s_add_u32 m0, %1, %4\n\ts_add_u32 %0, %2, %4\n\tbuffer_load_dwordx4 %3, %5, %0 offen lds"
                 : "=&s"(tmp)
                 : "s"(dwin), "s"(swin), "v"(r.voff), "i"(J * 4096), "s"(r.srd)
                 : "memory");
}
__device__ __forceinline__ void c_advance(CRing &r) {
    r.rd = (r.rd + C_WIN_BYTES) & (C_RING_BYTES - 1);
    r.wr = (r.wr + C_WIN_BYTES) & (C_RING_BYTES - 1);
    r.src += C_WIN_BYTES;
    r.src = r.src >= r.pass_bytes ? 0u : r.src;
}

typedef _Float16 half2c __attribute__((ext_vector_type(2)));

// activation of one accumulator pair -> two halves of the next layer's B operand
template <bool SOFTPLUS>
__device__ __forceinline__ half2c c_act_pair(float a0, float a1) {
    typedef float float2c __attribute__((ext_vector_type(2)));
    const half2c zs = __builtin_convertvector(float2c{a0, a1}, half2c);
    if constexpr (SOFTPLUS) {
        return softplus100_s16_pk(zs);
    } else {            // ReLU (test nets)
        return __builtin_elementwise_max(zs, half2c{(_Float16)0.f, (_Float16)0.f});
    }
}

// Positional encoding of one point, every column (embedder.py:21-31 order: x, sin f0 x, cos f0 x, sin f1 x, ...), x 16, on
// v_sin_f32 / v_cos_f32 (input in revolutions; |2^5 x| < 64 rad, absolute error ~1e-6 - the values are rounded to fp16)
__device__ __forceinline__ void c_encode(const float (&x)[3], float (&pe)[40]) {
    pe[0] = x[0] * A16_SCALE, pe[1] = x[1] * A16_SCALE, pe[2] = x[2] * A16_SCALE;
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float rev = x[c] * ((float)(1 << k) * 0.15915494309189535f);
            pe[3 + 6 * k + c] = __builtin_amdgcn_sinf(rev) * A16_SCALE;
            pe[3 + 6 * k + 3 + c] = __builtin_amdgcn_cosf(rev) * A16_SCALE;
        }
    pe[39] = 0.f;
}
// 8 consecutive columns c0 + 8 h .. of the encoding as one B-operand register set (h = lane >> 5); columns >= 39 are 0
template <int C0>
__device__ __forceinline__ half8 c_enc_step(const float (&pe)[40], bool h) {
    half8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c0 = C0 + i, c1 = C0 + 8 + i;
        const float v0 = (c0 >= 0 && c0 < 39) ? pe[c0 < 0 ? 39 : (c0 > 39 ? 39 : c0)] : 0.f;
        const float v1 = (c1 >= 0 && c1 < 39) ? pe[c1 < 0 ? 39 : (c1 > 39 ? 39 : c1)] : 0.f;
        r[i] = (_Float16)(h ? v1 : v0);
    }
    return r;
}

// ---- one window of a hidden layer = one 32-feature output tile: 8 groups of {DMA piece, 4 fragment reads for the next
// group, 4 MFMAs, one pair of the previous tile's epilogue}.  T: tile index; `cur` accumulates (preloaded with the bias),
// `prev` is the finished tile T - 1 (PREV: it exists), `nxt` receives tile T + 1's bias.
template <int T, int NT, bool PREV, bool SOFTPLUS>
__device__ __forceinline__ void c_hidden_tile(CRing &r, const char *ring, const float *bias_l,
                                              const half8 (&xin)[2 * NT], half8 (&xout)[2 * NT], half8 (&a)[2][4],
                                              f32x16 &cur, f32x16 &prev, f32x16 &nxt) {
    const unsigned dwin = r.lds0 + r.wr + r.wave_off, swin = r.src + r.wave_off;
    const char *rdp = ring + r.rd + r.voff;
    const char *rdn = ring + ((r.rd + C_WIN_BYTES) & (C_RING_BYTES - 1)) + r.voff;
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        if (J == 7) {
            // windows c + 1 (read from the next group on) must have landed: this wave's pieces of it were issued 2 windows
            // ago; outstanding may stay: windows c + 2, c + 3 so far = 8 + 7 pieces
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        switch (J) {
            case 0: c_dma<0>(r, dwin, swin); break;
            case 1: c_dma<1>(r, dwin, swin); break;
            case 2: c_dma<2>(r, dwin, swin); break;
            case 3: c_dma<3>(r, dwin, swin); break;
            case 4: c_dma<4>(r, dwin, swin); break;
            case 5: c_dma<5>(r, dwin, swin); break;
            case 6: c_dma<6>(r, dwin, swin); break;
            default: c_dma<7>(r, dwin, swin); break;
        }
        // next group's fragments
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[(J + 1) & 1][i] = J < 7 ? *reinterpret_cast<const half8 *>(rdp + (J + 1) * 4096 + i * 1024)
                                      : *reinterpret_cast<const half8 *>(rdn + i * 1024);
        if (J == 0 && T + 1 < NT) {        // bias of tile T + 1 -> its accumulator
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4v b = *reinterpret_cast<const float4v *>(bias_l + (T + 1) * 32 + 4 * q);
                nxt[4 * q] = b[0], nxt[4 * q + 1] = b[1], nxt[4 * q + 2] = b[2], nxt[4 * q + 3] = b[3];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[J & 1][i], xin[4 * J + i], cur, 0, 0, 0);
        if (PREV) {
            // pair J of tile T - 1: registers 2 J, 2 J + 1 -> half pair J & 3 of k-step 2 (T - 1) + (J >> 2)
            const half2c h2 = c_act_pair<SOFTPLUS>(prev[2 * J], prev[2 * J + 1]);
            half8 &o = xout[2 * (T - 1) + (J >> 2)];
            o[2 * (J & 3)] = h2[0];
            o[2 * (J & 3) + 1] = h2[1];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    c_advance(r);
}

// the epilogue of a layer's last tile (nothing left to hide it behind)
template <int NT, bool SOFTPLUS>
__device__ __forceinline__ void c_tail_epilogue(const f32x16 &prev, half8 (&xout)[2 * NT]) {
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        const half2c h2 = c_act_pair<SOFTPLUS>(prev[2 * J], prev[2 * J + 1]);
        half8 &o = xout[2 * (NT - 1) + (J >> 2)];
        o[2 * (J & 3)] = h2[0];
        o[2 * (J & 3) + 1] = h2[1];
    }
}

template <int NT, bool SOFTPLUS, int T = 0>
__device__ __forceinline__ void c_hidden_layer(CRing &r, const char *ring, const float *bias_l, const half8 (&xin)[2 * NT],
                                               half8 (&xout)[2 * NT], half8 (&a)[2][4], f32x16 (&acc)[3]) {
    if constexpr (T < NT) {
        c_hidden_tile<T, NT, (T > 0), SOFTPLUS>(r, ring, bias_l, xin, xout, a, acc[T % 3], acc[(T + 2) % 3],
                                                acc[(T + 1) % 3]);
        c_hidden_layer<NT, SOFTPLUS, T + 1>(r, ring, bias_l, xin, xout, a, acc);
    } else {
        c_tail_epilogue<NT, SOFTPLUS>(acc[(NT - 1) % 3], xout);
    }
}

// ---- layer 0: 8 tiles per window, 4 k-steps of the encoding each; every group is a whole tile with its own epilogue
template <int W, int NT, bool SOFTPLUS>
__device__ __forceinline__ void c_first_window(CRing &r, const char *ring, const float *bias_l, const half8 (&enc)[4],
                                               half8 (&xout)[2 * NT], half8 (&a)[2][4]) {
    const unsigned dwin = r.lds0 + r.wr + r.wave_off, swin = r.src + r.wave_off;
    const char *rdp = ring + r.rd + r.voff;
    const char *rdn = ring + ((r.rd + C_WIN_BYTES) & (C_RING_BYTES - 1)) + r.voff;
    f32x16 acc[2];
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        const int T = 8 * W + J;
        if (J == 7) {
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        switch (J) {
            case 0: c_dma<0>(r, dwin, swin); break;
            case 1: c_dma<1>(r, dwin, swin); break;
            case 2: c_dma<2>(r, dwin, swin); break;
            case 3: c_dma<3>(r, dwin, swin); break;
            case 4: c_dma<4>(r, dwin, swin); break;
            case 5: c_dma<5>(r, dwin, swin); break;
            case 6: c_dma<6>(r, dwin, swin); break;
            default: c_dma<7>(r, dwin, swin); break;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[(J + 1) & 1][i] = J < 7 ? *reinterpret_cast<const half8 *>(rdp + (J + 1) * 4096 + i * 1024)
                                      : *reinterpret_cast<const half8 *>(rdn + i * 1024);
        f32x16 &c = acc[J & 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v b = *reinterpret_cast<const float4v *>(bias_l + T * 32 + 4 * q);
            c[4 * q] = b[0], c[4 * q + 1] = b[1], c[4 * q + 2] = b[2], c[4 * q + 3] = b[3];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[J & 1][i], enc[i], c, 0, 0, 0);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const half2c h2 = c_act_pair<SOFTPLUS>(c[2 * p], c[2 * p + 1]);
            half8 &o = xout[2 * T + (p >> 2)];
            o[2 * (p & 3)] = h2[0];
            o[2 * (p & 3) + 1] = h2[1];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    c_advance(r);
}

// ---- last layer: one tile, no activation; row 0 of the tile is the SDF column
template <int NT>
__device__ __forceinline__ float c_last_window(CRing &r, const char *ring, const half8 (&xin)[2 * NT], half8 (&a)[2][4]) {
    const unsigned dwin = r.lds0 + r.wr + r.wave_off, swin = r.src + r.wave_off;
    const char *rdp = ring + r.rd + r.voff;
    const char *rdn = ring + ((r.rd + C_WIN_BYTES) & (C_RING_BYTES - 1)) + r.voff;
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    static_assert(2 * NT == 32, "one window = the 32 k-steps of a 512-wide layer");
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        if (J == 7) {
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        switch (J) {
            case 0: c_dma<0>(r, dwin, swin); break;
            case 1: c_dma<1>(r, dwin, swin); break;
            case 2: c_dma<2>(r, dwin, swin); break;
            case 3: c_dma<3>(r, dwin, swin); break;
            case 4: c_dma<4>(r, dwin, swin); break;
            case 5: c_dma<5>(r, dwin, swin); break;
            case 6: c_dma<6>(r, dwin, swin); break;
            default: c_dma<7>(r, dwin, swin); break;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[(J + 1) & 1][i] = J < 7 ? *reinterpret_cast<const half8 *>(rdp + (J + 1) * 4096 + i * 1024)
                                      : *reinterpret_cast<const half8 *>(rdn + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[J & 1][i], xin[4 * J + i], c, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    c_advance(r);
    return c[0];
}

// Start of a workgroup: the bias table, the first three windows of the stream, the first group's fragments.
template <int NT>
__device__ __forceinline__ void c_prime(const nefii_mlp &m, const CStream &g, size_t c_off_bytes, LdsC &lds, CRing &r,
                                        half8 (&a)[2][4]) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char *base = reinterpret_cast<const char *>(m.w_stream) + c_off_bytes;
    r.srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, g.frags * 1024, 0x00020000);
    r.voff = lane * 16;
    r.lds0 = (unsigned)reinterpret_cast<size_t>(lds.ring);
    r.rd = 0;
    r.pass_bytes = (unsigned)g.frags * 1024u;
    r.wave_off = wave * 1024;
    // bias table (plain loads: nothing else is in flight yet)
    const float4v *bsrc = reinterpret_cast<const float4v *>(base + (size_t)g.frags * 1024);
    float4v *bdst = reinterpret_cast<float4v *>(lds.bias);
    for (int i = threadIdx.x; i < g.bias_bytes / 16; i += 256) bdst[i] = bsrc[i];
    // windows 0, 1, 2
    r.wr = 0, r.src = 0;
    for (int w = 0; w < C_AHEAD_WINS; ++w) {
        const unsigned dwin = r.lds0 + r.wr + r.wave_off, swin = r.src + r.wave_off;
        c_dma<0>(r, dwin, swin), c_dma<1>(r, dwin, swin), c_dma<2>(r, dwin, swin), c_dma<3>(r, dwin, swin);
        c_dma<4>(r, dwin, swin), c_dma<5>(r, dwin, swin), c_dma<6>(r, dwin, swin), c_dma<7>(r, dwin, swin);
        r.wr = (r.wr + C_WIN_BYTES) & (C_RING_BYTES - 1);
        r.src += C_WIN_BYTES;
        r.src = r.src >= r.pass_bytes ? 0u : r.src;
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // window 0 landed (this wave's pieces)
    __syncthreads();                                           // ... everybody's, and the bias table
    const char *rdp = lds.ring + r.voff;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = *reinterpret_cast<const half8 *>(rdp + i * 1024);
}

// One pass: the wave's 32 queries (lane & 31; both halves of the wave hold the point) through the whole network.
// Returns the SDF of query lane & 31 (valid in every lane).
template <int HW, bool SOFTPLUS>
__device__ __forceinline__ float sdf_pass16c(const nefii_mlp &m, const CStream &g, LdsC &lds, CRing &r, half8 (&a)[2][4],
                                             const float (&x)[3]) {
    constexpr int NT = HW / 32;
    const bool h = (threadIdx.x & 32) != 0;
    const char *ring = lds.ring;
    half8 xin[2 * NT], xout[2 * NT];
    {
        float pe[40];
        c_encode(x, pe);
        half8 enc[4];
        enc[0] = c_enc_step<0>(pe, h), enc[1] = c_enc_step<16>(pe, h), enc[2] = c_enc_step<32>(pe, h), enc[3] = c_enc_step<48>(pe, h);
        const float *b0 = lds.bias + (h ? 16 : 0);
        c_first_window<0, NT, SOFTPLUS>(r, ring, b0, enc, xout, a);
        c_first_window<1, NT, SOFTPLUS>(r, ring, b0, enc, xout, a);
    }
    for (int l = 1; l < g.nh; ++l) {
#pragma unroll
        for (int s = 0; s < 2 * NT; ++s) xin[s] = xout[s];
        if (l == g.skip) {
            // the skip layer's input is cat[hidden (HW - 39), encoding (39)] (/ sqrt 2 folded into the weights): columns
            // HW - 39 .. HW - 1 = k-step 2 NT - 3 (upper half, from its second value on), 2 NT - 2, 2 NT - 1
            float pe[40];
            c_encode(x, pe);
            const half8 e0 = c_enc_step<-9>(pe, h), e1 = c_enc_step<7>(pe, h), e2 = c_enc_step<23>(pe, h);
            half8 &p = xin[2 * NT - 3];
#pragma unroll
            for (int i = 1; i < 8; ++i) p[i] = h ? e0[i] : p[i];
            xin[2 * NT - 2] = e1, xin[2 * NT - 1] = e2;
        }
        f32x16 acc[3];
        const float *bl = lds.bias + l * (NT * 32) + (h ? 16 : 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v b = *reinterpret_cast<const float4v *>(bl + 4 * q);
            acc[0][4 * q] = b[0], acc[0][4 * q + 1] = b[1], acc[0][4 * q + 2] = b[2], acc[0][4 * q + 3] = b[3];
        }
        c_hidden_layer<NT, SOFTPLUS>(r, ring, bl, xin, xout, a, acc);
    }
    const float s = c_last_window<NT>(r, ring, xout, a);
    // row 0 lives in register 0 of the lanes with h = 0; the upper half reads it from its partner lane
    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (threadIdx.x & 31), __builtin_bit_cast(int, s)));
    return s0 * (1.f / A16_SCALE) + m.layer[g.nh].bias[0];
}

}  // namespace nefii
