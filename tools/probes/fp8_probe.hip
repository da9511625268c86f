// Does the block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands) deliver its 2x-per-clock over the fp16
// form ON THIS POWER-LIMITED PART, on random operands - and at what clock?  (VERDICT r4 next #1b: the question to answer
// before the split evaluator's two correction products x_h w_l, x_l w_h are considered for it.)
// One wave per SIMD, 4 waves per workgroup, one workgroup per CU, bare MFMA loops, operands in registers:
//   f16  : v_mfma_f32_16x16x32_f16, CH independent accumulator chains (4 = nefii_mfma_sustained_probe's loop, 8)
//   fp8  : v_mfma_f32_16x16x32_fp8_fp8 (non-scaled: the fp16 form's rate per instruction)
//   mx8  : v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 x e4m3, unit scales (4x the K of the fp16 form per instruction)
//   mx4  : the same instruction on fp4 operands (e2m1), for the table's sake
// Cycles from s_memtime, time from HIP events; FLOPs = 2 x 16 x 16 x K per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/fp8_probe.hip -o tools/probes/fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lcg(unsigned &h) { h = h * 1664525u + 1013904223u; return h; }
// a random e4m3 byte with |x| in [2^-3, 2): exponent field 4..7, any mantissa, any sign (no NaN, no subnormal)
__device__ __forceinline__ unsigned rnd_e4m3x4(unsigned &h, int zero) {
    if (zero) return 0u;
    unsigned w = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned r = lcg(h) >> 8;
        w |= (((r & 0x80u)) | ((4u + ((r >> 8) & 3u)) << 3) | (r & 7u)) << (8 * i);
    }
    return w;
}

enum { K_F16 = 0, K_FP8 = 1, K_MX8 = 2, K_MX4 = 3, K_SPLIT = 4, K_MIX = 5 };

template <int KIND, int CH, int ZERO>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, int groups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    f32x4 acc[CH];
    for (int q = 0; q < CH; ++q)
        for (int i = 0; i < 4; ++i) acc[q][i] = 0.f;
    half8 a16[2], b16[CH];
    i32x8 a8[2], b8[CH];
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 8; ++j) a16[i][j] = ZERO ? (_Float16)0.f : (_Float16)((float)((lcg(h) >> 9) & 0xffff) * (2.f / 65536.f) - 1.f);
        for (int j = 0; j < 8; ++j) a8[i][j] = (int)rnd_e4m3x4(h, ZERO);
    }
    for (int q = 0; q < CH; ++q) {
        for (int j = 0; j < 8; ++j) b16[q][j] = ZERO ? (_Float16)0.f : (_Float16)((float)((lcg(h) >> 9) & 0xffff) * (2.f / 65536.f) - 1.f);
        for (int j = 0; j < 8; ++j) b8[q][j] = (int)rnd_e4m3x4(h, ZERO);
    }
    const int unit = 0x7f7f7f7f;          // block scales 2^0
    long a64[2], b64[CH];
    for (int i = 0; i < 2; ++i) a64[i] = ((long)a8[i][1] << 32) | (unsigned)a8[i][0];
    for (int q = 0; q < CH; ++q) b64[q] = ((long)b8[q][1] << 32) | (unsigned)b8[q][0];
    // inline assembly throughout: with the builtins hipcc 7.2 rotates the accumulators of such a loop through overlapping
    // register ranges (a[16:19] <- a[14:17], ...) and the false dependencies hold it at 31 cycles per instruction
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < groups; ++g) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                if (KIND == K_F16) {
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a16[i]), "v"(b16[q]));
                } else if (KIND == K_FP8) {
                    asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a64[i]), "v"(b64[q]));
                } else if (KIND == K_SPLIT) {    // one 128-deep chunk of a SPLIT product: 4 k-steps x (hh, hl, lh) on the fp16 form
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a16[i]), "v"(b16[q]));
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a16[i ^ 1]), "v"(b16[q]));
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a16[i]), "v"(b16[(q + 1) % CH]));
                    }
                } else if (KIND == K_MIX) {      // the same chunk with the two correction products on ONE block-scaled fp8 MFMA each
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a16[i]), "v"(b16[q]));
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[q]) : "v"(a8[i]), "v"(b8[q]), "v"(unit));
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[q]) : "v"(a8[i ^ 1]), "v"(b8[(q + 1) % CH]), "v"(unit));
                } else if (KIND == K_MX8) {      // both operands fp8 (e4m3): 8 registers each
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[q]) : "v"(a8[i]), "v"(b8[q]), "v"(unit));
                } else {                         // both operands fp4 (e2m1): 4 registers each
                    const i32x4 a4 = {a8[i][0], a8[i][1], a8[i][2], a8[i][3]}, b4 = {b8[q][0], b8[q][1], b8[q][2], b8[q][3]};
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                                 : "+v"(acc[q]) : "v"(a4), "v"(b4), "v"(unit));
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
    for (int q = 0; q < CH; ++q)
        for (int i = 0; i < 4; ++i) s += acc[q][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KIND, int CH, int ZERO>
void run(const char *name, float *out, unsigned long long *cyc, int groups) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<KIND, CH, ZERO>), dim3(256), dim3(256), 0, 0, out, cyc, groups);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, CH, ZERO>), dim3(256), dim3(256), 0, 0, out, cyc, groups);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), cyc, 8192, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto c : h) m += (double)c;
    m /= 1024.0;
    const double kdepth = KIND == K_F16 || KIND == K_FP8 ? 32.0 : 128.0;
    const double n_inst = 2.0 * CH * groups;
    if (KIND == K_SPLIT || KIND == K_MIX) {
        // n_inst = 128-deep chunks of one 16 x 16 split product; ALGORITHMIC flops: one product per chunk
        const double pf = 256.0 * 4 * n_inst * 2.0 * 16 * 16 * 128.0 / (ms * 1e-3) / 1e15;
        printf("%-64s %6.1f cycles per 128-deep chunk, %8.3f ms, clock %.2f GHz, %.3f PFLOP/s ALGORITHMIC chip-wide\n", name,
               m / n_inst, ms, m / (ms * 1e6), pf);
        return;
    }
    const double pf = 256.0 * 4 * n_inst * 2.0 * 16 * 16 * kdepth / (ms * 1e-3) / 1e15;
    printf("%-64s %6.1f cycles per instruction, %8.3f ms, clock %.2f GHz, %.2f PFLOP/s chip-wide\n", name, m / n_inst, ms,
           m / (ms * 1e6), pf);
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8192);
    const int G = 3712, GL = 250000;        // ~0.3 ms and ~25 ms of fp16 MFMAs
    run<K_F16, 4, 1>("f16 16x16x32, 4 chains, ZERO operands", out, cyc, G);
    run<K_F16, 4, 0>("f16 16x16x32, 4 chains, random operands", out, cyc, G);
    run<K_F16, 8, 0>("f16 16x16x32, 8 chains, random operands", out, cyc, G / 2);
    run<K_FP8, 4, 0>("fp8 16x16x32 (non-scaled), 4 chains, random operands", out, cyc, G);
    run<K_FP8, 8, 0>("fp8 16x16x32 (non-scaled), 8 chains, random operands", out, cyc, G / 2);
    run<K_MX8, 4, 1>("mx fp8 16x16x128 (scaled), 4 chains, ZERO operands", out, cyc, G / 2);
    run<K_MX8, 4, 0>("mx fp8 16x16x128 (scaled), 4 chains, random operands", out, cyc, G / 2);
    run<K_MX8, 8, 0>("mx fp8 16x16x128 (scaled), 8 chains, random operands", out, cyc, G / 4);
    run<K_MX4, 4, 0>("mx fp4 16x16x128 (scaled), 4 chains, random operands", out, cyc, G / 2);
    run<K_MX4, 8, 0>("mx fp4 16x16x128 (scaled), 8 chains, random operands", out, cyc, G / 4);
    printf("-- 25 ms launches (what the part SUSTAINS)\n");
    run<K_F16, 4, 0>("f16 16x16x32, 4 chains, random operands, long", out, cyc, GL);
    run<K_F16, 8, 0>("f16 16x16x32, 8 chains, random operands, long", out, cyc, GL / 2);
    run<K_MX8, 4, 0>("mx fp8 16x16x128 (scaled), 4 chains, random operands, long", out, cyc, GL / 2);
    run<K_MX8, 8, 0>("mx fp8 16x16x128 (scaled), 8 chains, random operands, long", out, cyc, GL / 4);
    printf("-- one 128-deep chunk of a split-precision product: 12 fp16 MFMAs (today) vs 4 fp16 + 2 block-scaled fp8 (round 6)\n");
    run<K_SPLIT, 4, 0>("split, 12 x f16 16x16x32 per chunk, 4 chains, random, long", out, cyc, GL / 12);
    run<K_MIX, 4, 0>("mix, 4 x f16 + 2 x mx fp8 per chunk, 4 chains, random, long", out, cyc, GL / 12);
    run<K_SPLIT, 8, 0>("split, 12 x f16 16x16x32 per chunk, 8 chains, random, long", out, cyc, GL / 24);
    run<K_MIX, 8, 0>("mix, 4 x f16 + 2 x mx fp8 per chunk, 8 chains, random, long", out, cyc, GL / 24);
    return 0;
}
