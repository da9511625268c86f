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
constexpr int C_RD_AHEAD = 3;                  // fragment reads run 3 groups (12 MFMAs) ahead of their use: 4 register sets

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

// Timing-only ablations (tools/ab_build.sh; results are wrong with any of them): NEFII_C_NO_DMA - no ring fill in the steady
// state; NEFII_C_NO_EPI - the activation is a bare convert; NEFII_C_NO_BARRIER - no s_waitcnt / s_barrier per window;
// NEFII_C_NO_MFMA - the matrix instructions replaced by an empty asm that keeps their operands alive.
//
// How the ring is filled (NEFII_C_FILL): 0 - LDS-DMA (buffer_load_dwordx4 ... lds): no registers, but a piece holds the
// issuing wave for ~66 cycles (profiles/r04/coarse_x_ablations.txt; MI355X_MICROARCH.md quotes ~60) - half of the 128 cycles
// its 4 MFMAs take, and a wave alone on its SIMD has nobody to issue MFMAs meanwhile; 1 (default) - through registers:
// buffer_load_dwordx4 into one of 8 stages now, ds_write_b128 of that stage one window (8 groups, > 1000 cycles) later.
#ifndef NEFII_C_FILL
#define NEFII_C_FILL 1
#endif
typedef unsigned uint4c __attribute__((ext_vector_type(4)));
struct CStages {
    uint4c s[8];
};

// one DMA piece: fragment 4 J + wave of the window being filled
template <int J>
__device__ __forceinline__ void c_dma(const CRing &r, unsigned dwin, unsigned swin) {
#ifdef NEFII_C_NO_DMA
    if (r.pass_bytes != 0xffffffffu) return;
#endif
    unsigned tmp;
    asm volatile("s_add_u32 m0, %1, %4\n\ts_add_u32 %0, %2, %4\n\tbuffer_load_dwordx4 %3, %5, %0 offen lds"
                 : "=&s"(tmp)
                 : "s"(dwin), "s"(swin), "v"(r.voff), "i"(J * 4096), "s"(r.srd)
                 : "memory", "scc");         // s_add writes SCC: hipcc had scheduled a compare / select pair around a piece
}
// group J's share of the fill.  DMA: piece J of window c + 3 -> its ring slot.  Registers: the piece fetched a window ago
// (piece J of window c + 2) -> its ring slot, then piece J of window c + 3 -> stage J.
// wdst: this lane's LDS address of fragment `wave` of group 0 of the slot being written
template <int J>
__device__ __forceinline__ void c_fill(const CRing &r, CStages &st, unsigned dwin, unsigned swin, char *wdst) {
#if NEFII_C_FILL == 0
    (void)st, (void)wdst;
    c_dma<J>(r, dwin, swin);
#else
    (void)dwin;
#ifdef NEFII_C_NO_DMA
    if (r.pass_bytes != 0xffffffffu) return;
#endif
    *reinterpret_cast<uint4c *>(wdst + J * 4096) = st.s[J];
    st.s[J] = __builtin_amdgcn_raw_buffer_load_b128(r.srd, r.voff, swin + J * 4096, 0);
#endif
}
// Called at the top of group 8 - C_RD_AHEAD of window c, the first group whose fragment reads (C_RD_AHEAD groups ahead) reach
// into window c + 1.  DMA fill: window c + 1 must have landed: this wave's pieces of it were issued 2 windows ago; what may
// stay outstanding is windows c + 2 and c + 3 so far = 8 + (8 - C_RD_AHEAD) pieces; the barrier makes it everybody's pieces -
// and tells that every wave is done with window c - 1, which the fill of window c + 3 has been overwriting since group 0
// (its group J only ever lands on fragments 4 J .. 4 J + 3, the ones group J of window c - 1 read a window ago).
// Register fill: window c + 1 was written (ds_write) during window c - 1 and every wave has since waited for younger LDS
// reads of its own, so the barrier alone publishes it; the slot written during window c + 1 is window c - 1's.
__device__ __forceinline__ void c_window_sync() {
#ifndef NEFII_C_NO_BARRIER
#if NEFII_C_FILL == 0
    static_assert(C_RD_AHEAD == 3, "vmcnt below = 8 + (8 - C_RD_AHEAD)");
    asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
}
// The matrix instruction in inline assembly, to pin its operands' register files: the accumulator in VGPRs (the epilogue's
// v_cvt_pk reads it directly; hipcc puts a builtin's accumulator into AGPRs and pays a v_accvgpr_read per value), the
// register-resident activations (B operand) in AGPRs (C_MFMA_A; nothing but the MFMAs ever reads them) or in VGPRs
// (C_MFMA_V: the encoding, the last layer's input), the weight fragments (A operand) in AGPRs too - ds_read_b128 writes them
// there directly - which leaves the 256 VGPRs to the output under construction (128), three accumulators (48), the fill's
// stages (32) and the epilogue.  hipcc does not know these are MFMAs and pads no hazard: every reader of
// an accumulator comes after a later MFMA or an explicit s_nop 15 (c_mfma_settle), operands are written hundreds of cycles
// before their MFMA.
#ifdef NEFII_C_NO_MFMA
#define C_MFMA_A(ACC, A, B) asm volatile("" : "+v"(ACC) : "a"(A), "a"(B))
#define C_MFMA_V(ACC, A, B) asm volatile("" : "+v"(ACC) : "a"(A), "v"(B))
#else
#define C_MFMA_A(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(A), "a"(B))
#define C_MFMA_V(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(A), "v"(B))
#endif
// (the accumulator as an operand: a plain VALU reader of it is then ordered behind these asm statements - hipcc is free to
// move register-only code across an asm volatile that does not mention its registers)
__device__ __forceinline__ void c_mfma_settle(f32x16 &acc) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc)); }
// behind the NEXT MFMA (which issued 32 cycles after the accumulator's last one) + 4 wait states: the 12 states hipcc pads
// between an 8-pass MFMA and a VALU reader of its result
__device__ __forceinline__ void c_after_mfma(f32x16 &acc) { asm volatile("s_nop 3" : "+v"(acc)); }
#define C_SLOT_END() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void c_advance(CRing &r) {
    r.rd = (r.rd + C_WIN_BYTES) & (C_RING_BYTES - 1);
    r.wr = (r.wr + C_WIN_BYTES) & (C_RING_BYTES - 1);
    r.src += C_WIN_BYTES;
    r.src = r.src >= r.pass_bytes ? 0u : r.src;
}

typedef _Float16 half2c __attribute__((ext_vector_type(2)));

// The activation of one accumulator pair (-> two halves of the next layer's B operand) in four stages, one per MFMA slot:
//   16 softplus(z) = max(zs, 0) + (16 ln2 / 100) log2(1 + e),  e = 2^(-|zs| 100 log2(e) / 16)  (mlp_tile.h, softplus100_s16_pk:
//   same polynomial), zs = 16 z = the accumulator (its bias went in as the initial value)
template <bool SOFTPLUS>
struct CAct {
    half2c zs, u, e, q;
    __device__ __forceinline__ void s0(float a0, float a1) {
        typedef float float2c __attribute__((ext_vector_type(2)));
        zs = __builtin_convertvector(float2c{a0, a1}, half2c);
#ifndef NEFII_C_NO_EPI
        if constexpr (SOFTPLUS) u = zs * (_Float16)(1.44269504088896340736f * 100.f / A16_SCALE);
#endif
    }
    __device__ __forceinline__ void s1() {
#ifndef NEFII_C_NO_EPI
        if constexpr (SOFTPLUS) {
            e[0] = __builtin_exp2f16(-__builtin_fabsf16(u[0]));
            e[1] = __builtin_exp2f16(-__builtin_fabsf16(u[1]));
        }
#endif
    }
    __device__ __forceinline__ void s2() {
#ifndef NEFII_C_NO_EPI
        if constexpr (SOFTPLUS) {
            constexpr float C_L = 0.69314718055994530942f * A16_SCALE / 100.f;
            const _Float16 B2 = (_Float16)(-0.67994519f * C_L), B3 = (_Float16)(0.32559803f * C_L), B4 = (_Float16)(-0.08477006f * C_L);
            q = __builtin_elementwise_fma(e, half2c{B4, B4}, half2c{B3, B3});
            q = __builtin_elementwise_fma(e, q, half2c{B2, B2});
        }
#endif
    }
    __device__ __forceinline__ half2c s3() {
#ifdef NEFII_C_NO_EPI
        return zs;
#else
        const half2c relu = __builtin_elementwise_max(zs, half2c{(_Float16)0.f, (_Float16)0.f});
        if constexpr (SOFTPLUS) {
            constexpr float C_L = 0.69314718055994530942f * A16_SCALE / 100.f;
            const _Float16 B1 = (_Float16)(1.43901483f * C_L);
            q = __builtin_elementwise_fma(e, q, half2c{B1, B1});
            return __builtin_elementwise_fma(e, q, relu);
        } else {
            return relu;            // ReLU (test nets)
        }
#endif
    }
};
// pair P (accumulator registers 2 P, 2 P + 1) of a finished tile -> half pair P & 3 of k-step 2 T + (P >> 2)
template <int NS>
__device__ __forceinline__ void c_store_pair(half8 (&xout)[NS], int T, int P, half2c v) {
    half8 &o = xout[2 * T + (P >> 2)];
    o[2 * (P & 3)] = v[0];
    o[2 * (P & 3) + 1] = v[1];
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

// ---- one window of a hidden layer = one 32-feature output tile = 8 groups of 4 MFMA slots.  A slot is one MFMA (32 cycles
// of the matrix pipe, 8 of them issue) and, behind it, its share of everything else - the order in the instruction stream is
// what lets an in-order wave run vector work in the MFMA's shadow (fillers lumped together just add to the MFMAs' time:
// the first version of this kernel, compiler-scheduled, took 308 cycles per group):
//   slot 0: group J's share of the ring fill; the previous tile's pair J: convert, scale
//   slot 1: the 4 fragment reads of group J + 1; pair J: the two exponentials
//   slot 2: (group 0) the next tile's bias -> its accumulator; pair J: polynomial
//   slot 3: pair J: polynomial, max, store
// T: tile index; `cur` accumulates (preloaded with the bias), `prev` is the finished tile T - 1 (PREV: it exists), `nxt`
// receives tile T + 1's bias.
template <int T, int NT, bool PREV, bool SOFTPLUS>
__device__ __forceinline__ void c_hidden_tile(CRing &r, CStages &st, char *ring, const float *bias_l,
                                              const half8 (&xin)[2 * NT], half8 (&xout)[2 * NT], half8 (&a)[4][4],
                                              f32x16 &cur, f32x16 &prev, f32x16 &nxt) {
    const unsigned dwin = r.lds0 + r.wr + r.wave_off, swin = r.src + r.wave_off;
    const char *rdp = ring + r.rd + r.voff;
    const char *rdn = ring + ((r.rd + C_WIN_BYTES) & (C_RING_BYTES - 1)) + r.voff;
    char *wdst = ring + r.wr + r.wave_off + r.voff;
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        CAct<SOFTPLUS> act;
        if (J == 8 - C_RD_AHEAD) c_window_sync();
        C_MFMA_A(cur, a[J & 3][0], xin[4 * J]);
        if (PREV && J == 0) c_after_mfma(prev);     // its readers come behind this tile's first MFMA: 32 cycles after its own last one
        switch (J) {
            case 0: c_fill<0>(r, st, dwin, swin, wdst); break;
            case 1: c_fill<1>(r, st, dwin, swin, wdst); break;
            case 2: c_fill<2>(r, st, dwin, swin, wdst); break;
            case 3: c_fill<3>(r, st, dwin, swin, wdst); break;
            case 4: c_fill<4>(r, st, dwin, swin, wdst); break;
            case 5: c_fill<5>(r, st, dwin, swin, wdst); break;
            case 6: c_fill<6>(r, st, dwin, swin, wdst); break;
            default: c_fill<7>(r, st, dwin, swin, wdst); break;
        }
        if (PREV) act.s0(prev[2 * J], prev[2 * J + 1]);
        C_SLOT_END();
        C_MFMA_A(cur, a[J & 3][1], xin[4 * J + 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[(J + C_RD_AHEAD) & 3][i] = J + C_RD_AHEAD < 8
                                             ? *reinterpret_cast<const half8 *>(rdp + (J + C_RD_AHEAD) * 4096 + i * 1024)
                                             : *reinterpret_cast<const half8 *>(rdn + (J + C_RD_AHEAD - 8) * 4096 + i * 1024);
        if (PREV) act.s1();
        C_SLOT_END();
        C_MFMA_A(cur, a[J & 3][2], xin[4 * J + 2]);
        if (J == 0 && T + 1 < NT) {        // bias of tile T + 1 -> its accumulator
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4v b = *reinterpret_cast<const float4v *>(bias_l + (T + 1) * 32 + 4 * q);
                nxt[4 * q] = b[0], nxt[4 * q + 1] = b[1], nxt[4 * q + 2] = b[2], nxt[4 * q + 3] = b[3];
            }
        }
        if (PREV) act.s2();
        C_SLOT_END();
        C_MFMA_A(cur, a[J & 3][3], xin[4 * J + 3]);
        if (PREV) c_store_pair(xout, T - 1, J, act.s3());
        C_SLOT_END();
    }
    c_advance(r);
}

// the epilogue of a layer's last tile (nothing left to hide it behind)
template <int NT, bool SOFTPLUS>
__device__ __forceinline__ void c_tail_epilogue(f32x16 &prev, half8 (&xout)[2 * NT]) {
    c_mfma_settle(prev);
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        CAct<SOFTPLUS> act;
        act.s0(prev[2 * J], prev[2 * J + 1]);
        act.s1();
        act.s2();
        c_store_pair(xout, NT - 1, J, act.s3());
    }
}

template <int NT, bool SOFTPLUS, int T = 0>
__device__ __forceinline__ void c_hidden_layer(CRing &r, CStages &st, char *ring, const float *bias_l,
                                               const half8 (&xin)[2 * NT], half8 (&xout)[2 * NT], half8 (&a)[4][4],
                                               f32x16 (&acc)[3]) {
    if constexpr (T < NT) {
        c_hidden_tile<T, NT, (T > 0), SOFTPLUS>(r, st, ring, bias_l, xin, xout, a, acc[T % 3], acc[(T + 2) % 3],
                                                acc[(T + 1) % 3]);
        c_hidden_layer<NT, SOFTPLUS, T + 1>(r, st, ring, bias_l, xin, xout, a, acc);
    } else {
        c_tail_epilogue<NT, SOFTPLUS>(acc[(NT - 1) % 3], xout);
    }
}

// ---- layer 0: 8 tiles per window, 4 k-steps of the encoding each: every group is a whole tile; its epilogue (8 pairs, two
// per slot) runs behind the next group's MFMAs.  acc[J & 1] accumulates tile 8 W + J, acc[(J + 1) & 1] is the finished one.
template <int W, int NT, bool SOFTPLUS>
__device__ __forceinline__ void c_first_window(CRing &r, CStages &st, char *ring, const float *bias_l, const half8 (&enc)[4],
                                               half8 (&xout)[2 * NT], half8 (&a)[4][4], f32x16 (&acc)[2]) {
    const unsigned dwin = r.lds0 + r.wr + r.wave_off, swin = r.src + r.wave_off;
    const char *rdp = ring + r.rd + r.voff;
    const char *rdn = ring + ((r.rd + C_WIN_BYTES) & (C_RING_BYTES - 1)) + r.voff;
    char *wdst = ring + r.wr + r.wave_off + r.voff;
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        const int T = 8 * W + J;
        f32x16 &c = acc[J & 1];
        f32x16 &p = acc[(J + 1) & 1];
        if (J == 8 - C_RD_AHEAD) c_window_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v b = *reinterpret_cast<const float4v *>(bias_l + T * 32 + 4 * q);
            c[4 * q] = b[0], c[4 * q + 1] = b[1], c[4 * q + 2] = b[2], c[4 * q + 3] = b[3];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            C_MFMA_V(c, a[J & 3][i], enc[i]);
            if (i == 0) {
                if (T > 0) c_after_mfma(p);
                switch (J) {
                    case 0: c_fill<0>(r, st, dwin, swin, wdst); break;
                    case 1: c_fill<1>(r, st, dwin, swin, wdst); break;
                    case 2: c_fill<2>(r, st, dwin, swin, wdst); break;
                    case 3: c_fill<3>(r, st, dwin, swin, wdst); break;
                    case 4: c_fill<4>(r, st, dwin, swin, wdst); break;
                    case 5: c_fill<5>(r, st, dwin, swin, wdst); break;
                    case 6: c_fill<6>(r, st, dwin, swin, wdst); break;
                    default: c_fill<7>(r, st, dwin, swin, wdst); break;
                }
            }
            if (i == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    a[(J + C_RD_AHEAD) & 3][k] = J + C_RD_AHEAD < 8
                                                     ? *reinterpret_cast<const half8 *>(rdp + (J + C_RD_AHEAD) * 4096 + k * 1024)
                                                     : *reinterpret_cast<const half8 *>(rdn + (J + C_RD_AHEAD - 8) * 4096 + k * 1024);
            }
            if (T > 0) {            // pairs 2 i, 2 i + 1 of tile T - 1
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    CAct<SOFTPLUS> act;
                    act.s0(p[4 * i + 2 * e], p[4 * i + 2 * e + 1]);
                    act.s1();
                    act.s2();
                    c_store_pair(xout, T - 1, 2 * i + e, act.s3());
                }
            }
            C_SLOT_END();
        }
    }
    c_advance(r);
}

// ---- last layer: one tile, no activation; row 0 of the tile is the SDF column
template <int NT>
__device__ __forceinline__ float c_last_window(CRing &r, CStages &st, char *ring, const half8 (&xin)[2 * NT],
                                               half8 (&a)[4][4]) {
    const unsigned dwin = r.lds0 + r.wr + r.wave_off, swin = r.src + r.wave_off;
    const char *rdp = ring + r.rd + r.voff;
    const char *rdn = ring + ((r.rd + C_WIN_BYTES) & (C_RING_BYTES - 1)) + r.voff;
    char *wdst = ring + r.wr + r.wave_off + r.voff;
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    static_assert(2 * NT == 32, "one window = the 32 k-steps of a 512-wide layer");
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        if (J == 8 - C_RD_AHEAD) c_window_sync();
        C_MFMA_V(c, a[J & 3][0], xin[4 * J]);
        switch (J) {
            case 0: c_fill<0>(r, st, dwin, swin, wdst); break;
            case 1: c_fill<1>(r, st, dwin, swin, wdst); break;
            case 2: c_fill<2>(r, st, dwin, swin, wdst); break;
            case 3: c_fill<3>(r, st, dwin, swin, wdst); break;
            case 4: c_fill<4>(r, st, dwin, swin, wdst); break;
            case 5: c_fill<5>(r, st, dwin, swin, wdst); break;
            case 6: c_fill<6>(r, st, dwin, swin, wdst); break;
            default: c_fill<7>(r, st, dwin, swin, wdst); break;
        }
        C_SLOT_END();
        C_MFMA_V(c, a[J & 3][1], xin[4 * J + 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[(J + C_RD_AHEAD) & 3][i] = J + C_RD_AHEAD < 8
                                             ? *reinterpret_cast<const half8 *>(rdp + (J + C_RD_AHEAD) * 4096 + i * 1024)
                                             : *reinterpret_cast<const half8 *>(rdn + (J + C_RD_AHEAD - 8) * 4096 + i * 1024);
        C_SLOT_END();
        C_MFMA_V(c, a[J & 3][2], xin[4 * J + 2]);
        C_MFMA_V(c, a[J & 3][3], xin[4 * J + 3]);
        C_SLOT_END();
    }
    c_advance(r);
    c_mfma_settle(c);
    return c[0];
}

// Start of a workgroup: the bias table, the first windows of the stream, the first group's fragments.
template <int NT>
__device__ __forceinline__ void c_prime(const nefii_mlp &m, const CStream &g, size_t c_off_bytes, LdsC &lds, CRing &r,
                                        CStages &st, half8 (&a)[4][4]) {
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
    r.wr = 0, r.src = 0;
#if NEFII_C_FILL == 0
    // windows 0, 1, 2 by DMA
    for (int w = 0; w < C_AHEAD_WINS; ++w) {
        const unsigned dwin = r.lds0 + r.wr + r.wave_off, swin = r.src + r.wave_off;
        c_dma<0>(r, dwin, swin), c_dma<1>(r, dwin, swin), c_dma<2>(r, dwin, swin), c_dma<3>(r, dwin, swin);
        c_dma<4>(r, dwin, swin), c_dma<5>(r, dwin, swin), c_dma<6>(r, dwin, swin), c_dma<7>(r, dwin, swin);
        r.wr = (r.wr + C_WIN_BYTES) & (C_RING_BYTES - 1);
        r.src += C_WIN_BYTES;
        r.src = r.src >= r.pass_bytes ? 0u : r.src;
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // window 0 landed (this wave's pieces)
#else
    // windows 0 and 1 into the ring, window 2 into the stages (written to the ring during window 0, the slot r.wr)
    for (int w = 0; w < C_AHEAD_WINS; ++w) {
        const unsigned swin = r.src + r.wave_off;
#pragma unroll
        for (int J = 0; J < 8; ++J) st.s[J] = __builtin_amdgcn_raw_buffer_load_b128(r.srd, r.voff, swin + J * 4096, 0);
        if (w + 1 < C_AHEAD_WINS) {
            char *wdst = lds.ring + r.wr + r.wave_off + r.voff;
#pragma unroll
            for (int J = 0; J < 8; ++J) *reinterpret_cast<uint4c *>(wdst + J * 4096) = st.s[J];
            r.wr = (r.wr + C_WIN_BYTES) & (C_RING_BYTES - 1);
        }
        r.src += C_WIN_BYTES;
        r.src = r.src >= r.pass_bytes ? 0u : r.src;
    }
#endif
    __syncthreads();                                           // everybody's pieces, and the bias table
    const char *rdp = lds.ring + r.voff;
#pragma unroll
    for (int gq = 0; gq < C_RD_AHEAD; ++gq)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[gq][i] = *reinterpret_cast<const half8 *>(rdp + gq * 4096 + i * 1024);
}

#ifdef NEFII_C_STAMPS
// stamp builds (tools/coarse_x_stamps.py): s_memtime at the layer boundaries of workgroup 0's second pass, per wave
__device__ unsigned long long g_c_stamps[4 * 16];
__device__ int g_c_stamp_pass;
#define C_STAMP(i)                                                                                         \
    if (blockIdx.x == 0 && g_c_stamp_pass == 1 && (threadIdx.x & 63) == 0)                                 \
        g_c_stamps[(threadIdx.x >> 6) * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define C_STAMP(i)
#endif
#ifdef NEFII_C_DEBUG
// debug builds (tools/coarse_x_debug.py): nefii_debug_c_select((layer + 1) << 10 | column) makes a pass return 16 x that layer's
// activation of that column instead of the SDF ((layer + 17) << 10 | column: the layer's input)
__device__ int g_c_dbg;
template <int NS>
__device__ __forceinline__ float c_pick(const half8 (&xo)[NS], int k) {
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (s == (k >> 4) && j == (k & 7)) v = (float)xo[s][j];
    const int src = (threadIdx.x & 31) + 32 * ((k >> 3) & 1);
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * src, __builtin_bit_cast(int, v)));
}
#endif

// One pass: the wave's 32 queries (lane & 31; both halves of the wave hold the point) through the whole network.
// Returns the SDF of query lane & 31 (valid in every lane).
template <int HW, bool SOFTPLUS>
__device__ __forceinline__ float sdf_pass16c(const nefii_mlp &m, const CStream &g, LdsC &lds, CRing &r, CStages &st,
                                             half8 (&a)[4][4], const float (&x)[3]) {
    constexpr int NT = HW / 32;
    const bool h = (threadIdx.x & 32) != 0;
    char *ring = lds.ring;
    half8 xin[2 * NT], xout[2 * NT];
    C_STAMP(0);
    {
        float pe[40];
        c_encode(x, pe);
        half8 enc[4];
        enc[0] = c_enc_step<0>(pe, h), enc[1] = c_enc_step<16>(pe, h), enc[2] = c_enc_step<32>(pe, h), enc[3] = c_enc_step<48>(pe, h);
        const float *b0 = lds.bias + (h ? 16 : 0);
        f32x16 acc0[2];
        c_first_window<0, NT, SOFTPLUS>(r, st, ring, b0, enc, xout, a, acc0);
        c_first_window<1, NT, SOFTPLUS>(r, st, ring, b0, enc, xout, a, acc0);
        c_mfma_settle(acc0[1]);
#pragma unroll
        for (int P = 0; P < 8; ++P) {           // tile NT - 1's epilogue
            CAct<SOFTPLUS> act;
            act.s0(acc0[1][2 * P], acc0[1][2 * P + 1]);
            act.s1();
            act.s2();
            c_store_pair(xout, NT - 1, P, act.s3());
        }
    }
#ifdef NEFII_C_DEBUG
    const int dbg = g_c_dbg;
    float dbg_val = 0.f;
    if ((dbg >> 10) == 1) dbg_val = c_pick<2 * NT>(xout, dbg & 1023);
#endif
    for (int l = 1; l < g.nh; ++l) {
        C_STAMP(l);
        if (l == g.skip) {
            // the skip layer's input is cat[hidden (HW - 39), encoding (39)] (/ sqrt 2 folded into the weights): columns
            // HW - 39 .. HW - 1 = k-step 2 NT - 3 (upper half, from its second value on), 2 NT - 2, 2 NT - 1
            float pe[40];
            c_encode(x, pe);
            const half8 e0 = c_enc_step<-9>(pe, h), e1 = c_enc_step<7>(pe, h), e2 = c_enc_step<23>(pe, h);
            half8 &p = xout[2 * NT - 3];
#pragma unroll
            for (int i = 1; i < 8; ++i) p[i] = h ? e0[i] : p[i];
            xout[2 * NT - 2] = e1, xout[2 * NT - 1] = e2;
        }
#pragma unroll
        for (int s = 0; s < 2 * NT; ++s) xin[s] = xout[s];     // -> AGPRs (the MFMAs' B operand class): 128 v_accvgpr_write
        f32x16 acc[3];
        const float *bl = lds.bias + l * (NT * 32) + (h ? 16 : 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v b = *reinterpret_cast<const float4v *>(bl + 4 * q);
            acc[0][4 * q] = b[0], acc[0][4 * q + 1] = b[1], acc[0][4 * q + 2] = b[2], acc[0][4 * q + 3] = b[3];
        }
        c_hidden_layer<NT, SOFTPLUS>(r, st, ring, bl, xin, xout, a, acc);
#ifdef NEFII_C_DEBUG
        if ((dbg >> 10) == l + 1) dbg_val = c_pick<2 * NT>(xout, dbg & 1023);
        if ((dbg >> 10) == l + 17) dbg_val = c_pick<2 * NT>(xin, dbg & 1023);     // the layer's INPUT (after the skip patch)
#endif
    }
    C_STAMP(g.nh);
    const float s = c_last_window<NT>(r, st, ring, xout, a);
    C_STAMP(g.nh + 1);
#ifdef NEFII_C_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) g_c_stamp_pass = g_c_stamp_pass + 1;
#endif
    // row 0 lives in register 0 of the lanes with h = 0; the upper half reads it from its partner lane
    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (threadIdx.x & 31), __builtin_bit_cast(int, s)));
#ifdef NEFII_C_DEBUG
    if (dbg) return dbg_val;
#endif
    return s0 * (1.f / A16_SCALE) + m.layer[g.nh].bias[0];
}

}  // namespace nefii
