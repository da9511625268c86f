// Do independent VALU instructions execute in the shadow of MFMAs on gfx950 - from the same wave, and from the SIMD's other wave?
// Inline-assembly loops (hipcc cannot rearrange them), every operand register distinct, no dependencies between the classes:
//   same   : each wave issues 8 independent v_mfma_f32_16x16x32_f16 per iteration with V vector instructions behind each
//   pair   : 8-wave workgroups - waves 0-3 (one per SIMD) run the bare MFMA loop, waves 4-7 a bare VALU loop of V x 8 per
//            iteration, the same number of iterations; each wave stamps its own s_memtime
// VT: 0 v_pk_fma_f16, 1 v_fma_f32, 2 v_exp_f16 (transcendental), 3 v_pk_mul_f16 with an SGPR-free constant pair
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/coexec_probe.hip -o tools/probes/coexec_probe
// Run:   tools/probes/coexec_probe [blocks (default 256; 1 = one CU, no power effect)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF(i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i) & 3]), "v"(b[(i) >> 2]))
template <int VT>
__device__ __forceinline__ void valu(unsigned &x, unsigned c1, unsigned c2) {
    if (VT == 0) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    if (VT == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    if (VT == 2) asm volatile("v_exp_f16 %0, %0" : "+v"(x));
    if (VT == 3) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x) : "v"(c1));
}

template <int V, int VT, bool DO_MFMA, bool DO_VALU>
__device__ __forceinline__ unsigned long long loop(int iters, float *sink) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
    half8 a[4], b[2];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.37f * (float)(((lane * 7 + i * 13 + j * 29) * 2654435761u >> 20) & 255) / 64.f - 0.7f);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)(0.21f * (float)(((lane * 11 + i * 17 + j * 31) * 2654435761u >> 20) & 255) / 64.f - 0.4f);
    unsigned x[32];
    for (int i = 0; i < 32; ++i) x[i] = 0x3c003800u + lane + i;        // packed halves near 1 / 0.5 (fp32: a small normal)
    const unsigned c1 = VT == 1 ? 0x3f7fff00u : 0x3bff3bffu, c2 = VT == 1 ? 0x33000000u : 0x0c000c00u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (DO_MFMA) MF(i);
            if (DO_VALU) {
#pragma unroll
                for (int q = 0; q < V; ++q) valu<VT>(x[(i * V + q) & 31], c1, c2);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 32; ++i) s += (float)x[i];
    *sink = s;
    return t1 - t0;
}

// MODE 0: 4 waves, every wave MFMA + V VALU per MFMA.  MODE 1: 8 waves, 0-3 MFMA only, 4-7 VALU only.
// MODE 2: 4 waves, VALU only (what V x 8 per iteration cost alone).  MODE 3: 8 waves, all of them MFMA + VALU.
template <int V, int VT, int MODE>
__global__ __launch_bounds__(512, 2) void k(float *out, unsigned long long *cyc, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned long long c;
    float s;
    if (MODE == 0 || MODE == 3)
        c = loop<V, VT, true, (V > 0)>(iters, &s);
    else if (MODE == 2)
        c = loop<V, VT, false, true>(iters, &s);
    else if (wave < 4)
        c = loop<V, VT, true, false>(iters, &s);
    else
        c = loop<V, VT, false, true>(iters, &s);
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = c;
}

static int g_blocks = 256;
template <int V, int VT, int MODE>
void run(const char *name) {
    const int waves = (MODE == 1 || MODE == 3) ? 8 : 4, iters = 20000;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, sizeof(float) * 512 * g_blocks);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * 8 * g_blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    std::vector<unsigned long long> h(8 * g_blocks);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<V, VT, MODE>), dim3(g_blocks), dim3(64 * waves), 0, 0, out, cyc, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 8 * g_blocks, hipMemcpyDeviceToHost);
    double cm = 0, cv = 0;
    for (int bk = 0; bk < g_blocks; ++bk)
        for (int w = 0; w < waves; ++w) (w < 4 ? cm : cv) += (double)h[bk * 8 + w];
    cm /= 4.0 * g_blocks * iters;
    if (waves == 8) cv /= 4.0 * g_blocks * iters;
    // s_memtime ticks at 100 MHz on this part: ticks x (clock / 100 MHz) = cycles; the clock comes from the bare MFMA loop's
    // known 16 cycles per instruction instead - report time per iteration and let the reader divide
    const double ns_iter = best * 1e6 / iters;
    printf("%-46s %8.2f ns / iteration of 8 MFMAs  (ticks / iteration: waves 0-3 %.3f%s", name, ns_iter, cm, waves == 8 ? "" : ")\n");
    if (waves == 8) printf(", waves 4-7 %.3f)\n", cv);
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main(int argc, char **argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    printf("blocks %d (one per CU up to 256), 20000 iterations; an iteration = 8 independent v_mfma_f32_16x16x32_f16 = 128 matrix-pipe cycles\n", g_blocks);
    run<0, 0, 0>("MFMA only, 1 wave / SIMD");
    run<0, 0, 3>("MFMA only, 2 waves / SIMD (per-wave iteration)");
    run<1, 0, 2>("v_pk_fma_f16 only, 8 / iteration");
    run<3, 0, 2>("v_pk_fma_f16 only, 24 / iteration");
    run<1, 0, 0>("same wave: MFMA + 1 v_pk_fma_f16 each");
    run<2, 0, 0>("same wave: MFMA + 2 v_pk_fma_f16 each");
    run<3, 0, 0>("same wave: MFMA + 3 v_pk_fma_f16 each");
    run<4, 0, 0>("same wave: MFMA + 4 v_pk_fma_f16 each");
    run<3, 1, 0>("same wave: MFMA + 3 v_fma_f32 each");
    run<1, 2, 0>("same wave: MFMA + 1 v_exp_f16 each");
    run<2, 2, 0>("same wave: MFMA + 2 v_exp_f16 each");
    run<1, 0, 1>("partner wave: 8 v_pk_fma_f16 / iteration");
    run<2, 0, 1>("partner wave: 16 v_pk_fma_f16 / iteration");
    run<3, 0, 1>("partner wave: 24 v_pk_fma_f16 / iteration");
    run<4, 0, 1>("partner wave: 32 v_pk_fma_f16 / iteration");
    run<3, 1, 1>("partner wave: 24 v_fma_f32 / iteration");
    run<1, 2, 1>("partner wave: 8 v_exp_f16 / iteration");
    run<2, 0, 3>("2 waves / SIMD, each MFMA + 2 v_pk_fma_f16");
    return 0;
}
