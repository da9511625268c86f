// TEST INFRASTRUCTURE (built into libnefii_canary.so by nefii_amd/build.py:build_canary; never loaded by nefii_amd).
// Packed-fp32 instruction FORMS as victims beside MFMA-streaming waves: every kernel runs one form
// in a long dependent chain per thread - written as inline assembly, so that the instruction and its modifiers are exactly
// what the name says - with the launch shape of the kernel that exposed the hazard (128 threads, 6 KB of LDS, < 80 VGPRs:
// one such wave fits beside two 209-register evaluator waves on a SIMD).  tools/concurrency_probe.py / the GPU suite compare
// the outputs bit for bit between an idle chip and a chip streaming MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float float2v __attribute__((ext_vector_type(2)));

// CLAIM: the kernel allocates 74 VGPRs like the kernel that exposed the hazard (exactly ONE such wave fits beside two
// 209-register evaluator waves, and it sits at the top of the SIMD's register file)
template <int FORM, bool CLAIM>
__global__ __launch_bounds__(128) void pk_form_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n, int iters) {
    __shared__ float pad[1536];
    if (CLAIM) asm volatile("" ::: "v73");
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = threadIdx.x; i < 1536; i += blockDim.x) pad[i] = 0.f;
    __syncthreads();
    if (p >= n) return;
    float2v a = {in[p * 6 + 0], in[p * 6 + 1]}, b = {in[p * 6 + 2], in[p * 6 + 3]}, acc = {in[p * 6 + 4], in[p * 6 + 5]};
    for (int it = 0; it < iters; ++it) {
        // acc <- acc * a + b (|a| < 1: the chain stays bounded), four dependent steps per trip
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
            if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
            if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
            if (FORM == 3) {
                float2v t;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(acc), "v"(a));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc) : "v"(t), "v"(b));
            }
            if (FORM == 4) {
                float2v t;
                asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(acc), "v"(a));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[1,0] neg_hi:[1,0]" : "=v"(acc) : "v"(t), "v"(b));
            }
            if (FORM == 5) {
                float2v t;
                asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(acc), "v"(acc));     // swap the halves
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(acc) : "v"(t), "v"(a), "v"(b));
            }
            // op_sel on a source's LOW result lane (both result lanes then read that source's HIGH register)
            if (FORM >= 10 && FORM <= 15) {
                float2v t;
                if (FORM == 10) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(acc), "v"(a));
                if (FORM == 11) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(acc), "v"(a));
                if (FORM == 12) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(t) : "v"(acc), "v"(a), "v"(b));
                if (FORM == 13) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(acc), "v"(a));
                if (FORM == 14) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(acc), "v"(a));
                if (FORM == 15) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(a), "v"(acc));
                if (FORM == 12) acc = t;
                else if (FORM == 13) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(acc) : "v"(t), "v"(b));
                else asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc) : "v"(t), "v"(b));
            }
            if (FORM == 6) {        // the same arithmetic without packed instructions (control)
                acc[0] = __builtin_fmaf(acc[0], a[0], b[0]);
                acc[1] = __builtin_fmaf(acc[1], a[1], b[1]);
                asm volatile("" : "+v"(acc));
            }
        }
    }
    out[p * 2 + 0] = acc[0] + pad[threadIdx.x];
    out[p * 2 + 1] = acc[1];
}

struct f3 { float x, y, z; };
__device__ inline f3 cross3(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline f3 axpy3(float s, f3 a, f3 b) { return {s * a.x + b.x, s * a.y + b.y, s * a.z + b.z}; }

// Compiler-formed packed fp32 (this file is compiled WITH the feature): the shapes of code the sampler has - a tangent
// frame from cross products and a change of basis (KIND 0: with the rsq normalisations, KIND 1: pure mul / add), and
// directions from sin / cos of the draws (KIND 2).
template <int KIND, bool CLAIM>
__global__ __launch_bounds__(128) void pk_natural_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n, int iters) {
    __shared__ float pad[1536];
    if (CLAIM) asm volatile("" ::: "v73");
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = threadIdx.x; i < 1536; i += blockDim.x) pad[i] = 0.f;
    __syncthreads();
    if (p >= n) return;
    const f3 n0 = {in[p * 6 + 0], in[p * 6 + 2], in[p * 6 + 3]};
    const f3 l = {in[p * 6 + 1] - 0.5f, in[p * 6 + 4] * 0.3f, in[p * 6 + 5] * 0.3f};
    f3 nn = n0, acc = {0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 2) {
            float sp, cp, st, ct;
            __sincosf(6.2831853f * (nn.x + l.x), &sp, &cp);
            __sincosf(3.1415926f * (nn.y + l.y), &st, &ct);
            const f3 d = {cp * st, sp * st, ct};
            const float a = d.x * l.x + d.y * l.y + d.z * l.z, b = d.x * nn.x + d.y * nn.y + d.z * nn.z;
            acc = axpy3(0.5f, acc, {a * d.x + b * l.x, a * d.y + b * l.y, a * d.z + b * l.z});
            nn = {0.5f * nn.x + 0.25f * d.z, 0.5f * nn.y + 0.25f * d.x, 0.5f * nn.z + 0.25f * d.y};
        } else {
            const f3 up = fabsf(nn.z) < 0.7f ? f3{0.f, 0.f, 1.f} : f3{1.f, 0.f, 0.f};
            f3 t = cross3(up, nn);
            if (KIND == 0) { const float r = rsqrtf(t.x * t.x + t.y * t.y + t.z * t.z + 1e-12f); t = {t.x * r, t.y * r, t.z * r}; }
            const f3 s = cross3(nn, t);
            f3 w = {l.x * t.x + l.y * s.x + l.z * nn.x, l.x * t.y + l.y * s.y + l.z * nn.y, l.x * t.z + l.y * s.z + l.z * nn.z};
            acc = axpy3(0.5f, acc, w);
            w = axpy3(0.5f, w, {0.5f * n0.x, 0.5f * n0.y, 0.5f * n0.z});
            if (KIND == 0) { const float r = rsqrtf(w.x * w.x + w.y * w.y + w.z * w.z + 1e-12f); w = {w.x * r, w.y * r, w.z * r}; }
            nn = w;
        }
    }
    out[p * 2 + 0] = acc.x + acc.z + pad[threadIdx.x];
    out[p * 2 + 1] = acc.y + nn.x;
}

// A transcendental instruction next to packed fp32, one relation per SEQ, with explicit registers (v40-v45) so that the
// dependencies are exactly the ones named:
//   0  WAR: v_pk_mul_f32 overwrites the pair that holds v_rsq_f32's SOURCE in the very next instruction
//   1  WAR control: a plain v_mul_f32 overwrites the source instead
//   2  RAW: v_pk_mul_f32 consumes v_rsq_f32's RESULT after the one wait state the compiler leaves (s_nop 0)
//   3  RAW with no wait state at all
//   4  an independent v_pk_fma_f32 between v_rsq_f32 and the packed consumer of its result
//   5  RAW control: a plain v_mul_f32 consumes the result after s_nop 0
//   6  WAW-ish: v_pk_mov_b32 writes the pair whose low half v_rsq_f32 is writing, then restores by a second v_rsq_f32
//   7  FOUR transcendentals back to back (cos, sin, sin, cos), then a v_pk_mul_f32 whose operand pairs they wrote
//   8  the same with two plain v_mul_f32 as consumers (control)
//   9  as 7 with three independent plain instructions before the packed consumer
//  10  as 7, the packed consumer reads only the LAST two results (one pair)
template <int SEQ>
__global__ __launch_bounds__(128) void pk_trans_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n, int iters) {
    __shared__ float pad[1536];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = threadIdx.x; i < 1536; i += blockDim.x) pad[i] = 0.f;
    __syncthreads();
    if (p >= n) return;
    float2v a = {in[p * 6 + 0], in[p * 6 + 1]}, b = {in[p * 6 + 2], in[p * 6 + 3]}, acc = {in[p * 6 + 4], in[p * 6 + 5]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float x = __builtin_fmaf(acc[0], acc[0], 2.f), y = __builtin_fmaf(acc[1], acc[1], 3.f);
            float r, p0, p1;
            if (SEQ == 0)
                asm volatile("v_mov_b32 v40, %3\n v_mov_b32 v41, %4\n v_rsq_f32 v42, v40\n v_pk_mul_f32 v[40:41], %5, %6\n s_nop 7\n"
                             "v_mov_b32 %0, v42\n v_mov_b32 %1, v40\n v_mov_b32 %2, v41"
                             : "=&v"(r), "=&v"(p0), "=&v"(p1) : "v"(x), "v"(y), "v"(a), "v"(b) : "v40", "v41", "v42", "v43");
            if (SEQ == 1)
                asm volatile("v_mov_b32 v40, %3\n v_mov_b32 v41, %4\n v_rsq_f32 v42, v40\n v_mul_f32 v40, %5, %6\n s_nop 7\n"
                             "v_mov_b32 %0, v42\n v_mov_b32 %1, v40\n v_mov_b32 %2, v41"
                             : "=&v"(r), "=&v"(p0), "=&v"(p1) : "v"(x), "v"(y), "v"(a[0]), "v"(b[0]) : "v40", "v41", "v42", "v43");
#define RAW_SEQ(MIDDLE)                                                                                           \
    asm volatile("v_mov_b32 v40, %[x]\n v_mov_b32 v43, %[y]\n v_mov_b32 v42, %[y]\n v_rsq_f32 v42, v40\n" MIDDLE           \
                 "v_pk_mul_f32 v[44:45], v[42:43], %[a]\n s_nop 7\n"                                               \
                 "v_mov_b32 %[r], v42\n v_mov_b32 %[p0], v44\n v_mov_b32 %[p1], v45"                                \
                 : [r] "=&v"(r), [p0] "=&v"(p0), [p1] "=&v"(p1), [acc] "+v"(acc)                                     \
                 : [x] "v"(x), [y] "v"(y), [a] "v"(a), [b] "v"(b) : "v40", "v41", "v42", "v43", "v44", "v45")
            if (SEQ == 2) RAW_SEQ("s_nop 0\n");
            if (SEQ == 3) RAW_SEQ("");
            if (SEQ == 4) RAW_SEQ("v_pk_fma_f32 %[acc], %[acc], %[a], %[b]\n");
            if (SEQ == 5)
                asm volatile("v_mov_b32 v40, %3\n v_mov_b32 v43, %4\n v_rsq_f32 v42, v40\n s_nop 0\n v_mul_f32 v44, v42, %5\n"
                             "v_mul_f32 v45, v43, %5\n s_nop 7\n v_mov_b32 %0, v42\n v_mov_b32 %1, v44\n v_mov_b32 %2, v45"
                             : "=&v"(r), "=&v"(p0), "=&v"(p1) : "v"(x), "v"(y), "v"(a[0]) : "v40", "v41", "v42", "v43", "v44", "v45");
            if (SEQ == 6)
                asm volatile("v_mov_b32 v40, %3\n v_mov_b32 v41, %4\n v_rsq_f32 v42, v40\n v_pk_mov_b32 v[42:43], %5, %6 op_sel:[1,0]\n"
                             "v_rsq_f32 v42, v41\n s_nop 7\n v_mov_b32 %0, v42\n v_mov_b32 %1, v43\n v_mov_b32 %2, v41"
                             : "=&v"(r), "=&v"(p0), "=&v"(p1) : "v"(x), "v"(y), "v"(a), "v"(b) : "v40", "v41", "v42", "v43");
            // several transcendental results in flight when the packed consumer issues (the shape hipcc gives sin / cos code)
#define QUEUE_SEQ(CONSUMER)                                                                                       \
    asm volatile("v_mul_f32 v40, 0x3a83126f, %[x]\n v_mul_f32 v41, 0x3a83126f, %[y]\n"                                \
                 "v_cos_f32 v46, v40\n v_sin_f32 v42, v40\n v_sin_f32 v45, v41\n v_cos_f32 v44, v41\n v_mov_b32 v43, v40\n"  \
                 CONSUMER "s_nop 7\n v_mov_b32 %[r], v46\n v_mov_b32 %[p0], v48\n v_mov_b32 %[p1], v49"               \
                 : [r] "=&v"(r), [p0] "=&v"(p0), [p1] "=&v"(p1)                                                      \
                 : [x] "v"(x), [y] "v"(y), [a] "v"(a), [b] "v"(b)                                                    \
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49")
            if (SEQ == 7) QUEUE_SEQ("v_pk_mul_f32 v[48:49], v[44:45], v[42:43] op_sel_hi:[1,0]\n");
            if (SEQ == 8) QUEUE_SEQ("v_mul_f32 v48, v44, v42\n v_mul_f32 v49, v45, v42\n");
            if (SEQ == 9) QUEUE_SEQ("v_mul_f32 v47, 0.5, v40\n v_mul_f32 v47, 0.5, v47\n v_mul_f32 v47, 0.5, v47\n"
                                    "v_pk_mul_f32 v[48:49], v[44:45], v[42:43] op_sel_hi:[1,0]\n");
            if (SEQ == 10) QUEUE_SEQ("v_pk_mul_f32 v[48:49], v[44:45], %[a]\n");
            acc[0] = __builtin_fmaf(acc[0], a[0], b[0] + 0.01f * r);
            acc[1] = __builtin_fmaf(acc[1], a[1], b[1] + 0.01f * (p0 + p1));
        }
    }
    out[p * 2 + 0] = acc[0] + pad[threadIdx.x];
    out[p * 2 + 1] = acc[1];
}

#define LAUNCH_TRANS(F, K) case F: hipLaunchKernelGGL((pk_trans_kernel<K>), grid, block, 0, st, in, out, n, iters); break;
#define LAUNCH_FORM(F)                                                                              \
    case F: if (claim) hipLaunchKernelGGL((pk_form_kernel<F, true>), grid, block, 0, st, in, out, n, iters);  \
            else hipLaunchKernelGGL((pk_form_kernel<F, false>), grid, block, 0, st, in, out, n, iters);       \
            break;
#define LAUNCH_NATURAL(F, K)                                                                        \
    case F: if (claim) hipLaunchKernelGGL((pk_natural_kernel<K, true>), grid, block, 0, st, in, out, n, iters); \
            else hipLaunchKernelGGL((pk_natural_kernel<K, false>), grid, block, 0, st, in, out, n, iters);      \
            break;

// form: 0-6, 10-15 the inline-assembly forms above, 7-9 the compiler-formed kinds 0-2; + 16: with the 74-VGPR allocation;
// 32-42: the transcendental sequences 0-10
extern "C" int nefii_canary_pk_form(int form, const float *in, float *out, int64_t n, int iters, void *stream) {
    const dim3 grid((unsigned)((n + 127) / 128)), block(128);
    hipStream_t st = (hipStream_t)stream;
    const bool claim = (form & 16) != 0;
    if (form >= 32) switch (form) {
        LAUNCH_TRANS(32, 0) LAUNCH_TRANS(33, 1) LAUNCH_TRANS(34, 2) LAUNCH_TRANS(35, 3) LAUNCH_TRANS(36, 4) LAUNCH_TRANS(37, 5)
        LAUNCH_TRANS(38, 6) LAUNCH_TRANS(39, 7) LAUNCH_TRANS(40, 8) LAUNCH_TRANS(41, 9) LAUNCH_TRANS(42, 10)
        default: return -1;
    }
    else switch (form & 15) {
        LAUNCH_FORM(0) LAUNCH_FORM(1) LAUNCH_FORM(2) LAUNCH_FORM(3) LAUNCH_FORM(4) LAUNCH_FORM(5) LAUNCH_FORM(6)
        LAUNCH_NATURAL(7, 0) LAUNCH_NATURAL(8, 1) LAUNCH_NATURAL(9, 2)
        LAUNCH_FORM(10) LAUNCH_FORM(11) LAUNCH_FORM(12) LAUNCH_FORM(13) LAUNCH_FORM(14) LAUNCH_FORM(15)
        default: return -1;
    }
    return (int)hipGetLastError();
}

// Synthetic NEIGHBOURS: 8-wave workgroups whose waves allocate 216 VGPRs (two per SIMD, 80 registers left - the shape of the
// single-pass evaluator) and execute ONE kind of instruction in a long loop, to find which of them disturbs the victim form.
//   0 v_mfma_f32_16x16x32_f16   1 v_pk_fma_f16   2 v_pk_fma_f32 (no op_sel)   3 v_fma_f32   4 ds_read_b128
//   5 v_exp_f16 (transcendental)   6 nothing but s_nop (the waves only sit there)   7 v_pk_fma_f16 + v_mfma interleaved
//   8-15 the packed-fp16 / SDWA / cross-lane forms of the single-pass evaluator's epilogue (SGPR sources, op_sel_hi)
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void neighbour_kernel(float *__restrict__ sink, int iters) {
    __shared__ float4v lds[512];
    asm volatile("" ::: "v208");
    const int t = threadIdx.x;
    lds[t] = float4v{1.f * t, 2.f, 3.f, 4.f};
    __syncthreads();
    half8v a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (t + i)); b[i] = (_Float16)(0.002f * (t - i)); }
    float4v c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    half2v h0 = {a[0], a[1]}, h1 = {b[0], b[1]}, h2 = {a[2], b[2]};
    float2v f0 = {0.5f, 0.25f}, f1 = {0.999f, 0.998f}, f2 = {0.001f, 0.002f};
    float s0 = 0.3f, s1 = 0.999f, s2 = 0.001f;
    const unsigned su = __builtin_amdgcn_readfirstlane(0x3bff3c00u + (unsigned)(iters & 1));       // (1.0, 0.9995) as fp16, in an SGPR
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0 || KIND == 7) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
            }
            if (KIND == 1 || KIND == 7) {
                asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(h0) : "v"(h1), "v"(h2));
                asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(h2) : "v"(h1), "v"(h0));
            }
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
            if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s0) : "v"(s1), "v"(s2));
            if (KIND == 4) { float4v r = lds[(t + it + u) & 511]; c0 += r; }
            if (KIND == 5) asm volatile("v_exp_f16 %0, %0" : "+v"(h0));
            if (KIND == 6) asm volatile("s_nop 7");
            if (KIND == 8) asm volatile("v_pk_mul_f16 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(h0) : "s"(su));
            if (KIND == 9) asm volatile("v_pk_mul_f16 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(h0) : "v"(h1));
            if (KIND == 10) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(h0) : "v"(h1), "s"(su));
            if (KIND == 11) asm volatile("v_pk_max_f16 %0, %0, 0" : "+v"(h0));
            if (KIND == 12) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(s0) : "v"(t * 4 ^ 64));
            if (KIND == 13) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(h0));
            if (KIND == 14) asm volatile("v_pack_b32_f16 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %2, %2" : "+v"(h0), "+v"(h1) : "v"(s0));
            if (KIND == 15) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(h0) : "s"(su), "v"(h1));
        }
    }
    if (sink != nullptr && iters < 0)
        sink[t] = c0[0] + c1[1] + c2[2] + c3[3] + (float)h0[0] + (float)h2[1] + f0[0] + f0[1] + s0;
}

#define LAUNCH_NEIGHBOUR(K) case K: hipLaunchKernelGGL((neighbour_kernel<K>), dim3(n_wg), dim3(512), 0, st, sink, iters); break;
extern "C" int nefii_canary_neighbour(int kind, float *sink, int iters, int n_wg, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
        LAUNCH_NEIGHBOUR(0) LAUNCH_NEIGHBOUR(1) LAUNCH_NEIGHBOUR(2) LAUNCH_NEIGHBOUR(3) LAUNCH_NEIGHBOUR(4) LAUNCH_NEIGHBOUR(5)
        LAUNCH_NEIGHBOUR(6) LAUNCH_NEIGHBOUR(7) LAUNCH_NEIGHBOUR(8) LAUNCH_NEIGHBOUR(9) LAUNCH_NEIGHBOUR(10) LAUNCH_NEIGHBOUR(11)
        LAUNCH_NEIGHBOUR(12) LAUNCH_NEIGHBOUR(13) LAUNCH_NEIGHBOUR(14) LAUNCH_NEIGHBOUR(15)
        default: return -1;
    }
    return (int)hipGetLastError();
}
