// What does a one-wave-per-SIMD MFMA stream pay for its fillers on gfx950?  Cycles (s_memtime) per group of 4 dependent
// v_mfma_f32_32x32x16_f16 with, per group: R ds_read_b128 (1 KiB each, 3 groups ahead), W ds_write_b128, G buffer_load_dwordx4,
// V packed-fp16 VALU instructions, T transcendentals - 4 waves per workgroup, one workgroup per CU, every CU busy.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/slot_probe.hip -o tools/probes/slot_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned uint4c __attribute__((ext_vector_type(4)));

template <int R, int W, int G, int V, int T, int ACC_AGPR, int A_VGPR, int UNR = 1, int RND = 0>
__global__ __launch_bounds__(256, 1) void k(const char *src, float *out, unsigned long long *cyc, int groups) {
    __shared__ char lds[144 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 36 * 1024; i += 256)
        reinterpret_cast<unsigned *>(lds)[i] = RND ? ((i * 2654435761u) & 0x3fff3fffu) ^ 0x80000000u * (i & 1) : 0x3c003c00u + (i & 7);
    __syncthreads();
    __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(src), 0, 4 << 20, 0x00020000);
    f32x16 acc, acc2;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f, acc2[i] = 0.f;
    half8 b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = RND ? (_Float16)(0.37f * (float)(((lane * 7 + i * 13 + j * 29) * 2654435761u >> 20) & 255) / 64.f - 0.7f) : (_Float16)(0.001f * (lane + i + j));
    half8 a[4][4];
    const char *rp = lds + lane * 16;
    for (int g = 0; g < 3; ++g)
        for (int i = 0; i < 4; ++i) a[g][i] = *reinterpret_cast<const half8 *>(rp + g * 4096 + i * 1024);
    uint4c st[8];
    for (int i = 0; i < 8; ++i) st[i] = uint4c{0, 0, 0, 0};
    half2 v = {(_Float16)0.5f, (_Float16)0.25f}, vv = {(_Float16)0.0001f, (_Float16)1.0f};
    unsigned off = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int g8 = 0; g8 < groups; g8 += 8 * UNR) {
#pragma unroll
      for (int U = 0; U < UNR; ++U)
#pragma unroll
        for (int J = 0; J < 8; ++J) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (ACC_AGPR) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[J & 3][i], b[i], acc, 0, 0, 0);
                } else if (A_VGPR) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[J & 3][i]), "v"(b[i]));
                } else {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a[J & 3][i]), "a"(b[i]));
                }
                if (i == 0) {
                    if (W) *reinterpret_cast<uint4c *>(lds + 65536 + ((off + J * 4096 + wave * 1024) & 0xffff) + lane * 16) = st[J];
                    if (G) st[J] = __builtin_amdgcn_raw_buffer_load_b128(srd, lane * 16, (off + J * 4096 + wave * 1024) & 0x3fffff, 0);
                }
                if (i == 1) {
#pragma unroll
                    for (int q = 0; q < R; ++q)
                        a[(J + 3) & 3][q] = *reinterpret_cast<const half8 *>(rp + ((J + 3) & 7) * 4096 + q * 1024);
                }
#pragma unroll
                for (int q = 0; q < (V + 3 - i) / 4; ++q) v = __builtin_elementwise_fma(v, vv, vv);
                if (i >= 2 && T) {
                    v[0] = __builtin_exp2f16(-__builtin_fabsf16(v[0]));
                    if (T > 1 || i == 3) v[1] = __builtin_exp2f16(-__builtin_fabsf16(v[1]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        off += 32768;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (!ACC_AGPR) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
    float s = (float)v[0] + (float)v[1];
    for (int i = 0; i < 8; ++i) s += (float)st[i][0];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same FLOPs per group on v_mfma_f32_16x16x32_f16: 8 instructions, QT accumulator chains (QT = 4: the shape of the "16s" / "16q"
// tiles, one weight fragment against 4 query tiles), R fragment reads per group
template <int R, int RND, int ZERO>
__global__ __launch_bounds__(256, 1) void k16(float *out, unsigned long long *cyc, int groups) {
    __shared__ char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16 * 1024; i += 256)
        reinterpret_cast<unsigned *>(lds)[i] = ZERO ? 0u : RND ? ((i * 2654435761u) & 0x3fff3fffu) ^ 0x80000000u * (i & 1) : 0x3c003c00u + (i & 7);
    __syncthreads();
    f32x4 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 4; ++i) acc[q][i] = 0.f;
    half8 b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j)
            b[i][j] = ZERO ? (_Float16)0.f : RND ? (_Float16)(0.37f * (float)(((lane * 7 + i * 13 + j * 29) * 2654435761u >> 20) & 255) / 64.f - 0.7f) : (_Float16)(0.001f * (lane + i + j));
    half8 a[4][2];
    const char *rp = lds + lane * 16;
    for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 2; ++i) a[g][i] = *reinterpret_cast<const half8 *>(rp + g * 2048 + i * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int g8 = 0; g8 < groups; g8 += 8) {
#pragma unroll
        for (int J = 0; J < 8; ++J) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[J & 3][i], b[q], acc[q], 0, 0, 0);
                if (R && i == 0) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) a[(J + 3) & 3][q] = *reinterpret_cast<const half8 *>(rp + ((J + 3) & 7) * 2048 + q * 1024);
                }
                if (R > 2 && i == 1) {      // the activation fragments' traffic: 2 more reads, results dropped into b (same values)
                    b[J & 3] = *reinterpret_cast<const half8 *>(rp + 32768 + (J & 3) * 1024);
                    b[(J + 1) & 3] = *reinterpret_cast<const half8 *>(rp + 32768 + ((J + 1) & 3) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 4; ++i) s += acc[q][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int R, int RND, int ZERO>
void run16(const char *name, float *out, unsigned long long *cyc) {
    const int groups = 128 * 29;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k16<R, RND, ZERO>), dim3(256), dim3(256), 0, 0, out, cyc, groups);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k16<R, RND, ZERO>), dim3(256), dim3(256), 0, 0, out, cyc, groups);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), cyc, 8192, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto c : h) m += (double)c;
    m /= 1024.0;
    const double pf = 256.0 * 4 * groups * 8 * 16384.0 / (ms * 1e-3) / 1e15;
    printf("%-58s %7.1f cycles per group of 8 MFMAs (%.3f ms, clock %.2f GHz, %.2f PFLOP/s chip-wide)\n", name, m / groups, ms, m / (ms * 1e6), pf);
}

template <int R, int W, int G, int V, int T, int ACC_AGPR, int A_VGPR, int UNR = 1, int RND = 0>
void run(const char *name, const char *src, float *out, unsigned long long *cyc) {
    const int groups = 128 * 29;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<R, W, G, V, T, ACC_AGPR, A_VGPR, UNR, RND>), dim3(256), dim3(256), 0, 0, src, out, cyc, groups);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, W, G, V, T, ACC_AGPR, A_VGPR, UNR, RND>), dim3(256), dim3(256), 0, 0, src, out, cyc, groups);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), cyc, 8192, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto c : h) m += (double)c;
    m /= 1024.0;
    const double pf = 256.0 * 4 * groups * 4 * 32768.0 / (ms * 1e-3) / 1e15;
    printf("%-58s %7.1f cycles per group of 4 MFMAs (%.3f ms, clock %.2f GHz, %.2f PFLOP/s chip-wide)\n", name, m / groups, ms, m / (ms * 1e6), pf);
}
int main() {
    char *src; float *out; unsigned long long *cyc;
    hipMalloc(&src, 4 << 20); hipMemset(src, 0, 4 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8192);
    //   R  W  G   V  T  accAGPR A_VGPR
    run<0, 0, 0,  0, 0, 0, 0>("MFMA only (acc VGPR, A/B AGPR)", src, out, cyc);
    run<0, 0, 0,  0, 0, 1, 0>("MFMA only (builtin: acc AGPR)", src, out, cyc);
    run<0, 0, 0,  0, 0, 0, 1>("MFMA only (acc VGPR, A/B VGPR)", src, out, cyc);
    run<4, 0, 0,  0, 0, 0, 0>("+ 4 ds_read_b128 -> AGPR", src, out, cyc);
    run<4, 0, 0,  0, 0, 0, 1>("+ 4 ds_read_b128 -> VGPR", src, out, cyc);
    run<4, 0, 0,  0, 0, 1, 0>("+ 4 ds_read_b128 (builtin MFMA)", src, out, cyc);
    run<2, 0, 0,  0, 0, 0, 0>("+ 2 ds_read_b128 -> AGPR", src, out, cyc);
    run<4, 1, 0,  0, 0, 0, 0>("+ 4 reads + 1 ds_write_b128", src, out, cyc);
    run<4, 0, 1,  0, 0, 0, 0>("+ 4 reads + 1 buffer_load_dwordx4", src, out, cyc);
    run<4, 1, 1,  0, 0, 0, 0>("+ 4 reads + write + load", src, out, cyc);
    run<0, 0, 0,  8, 0, 0, 0>("MFMA + 8 pk_fma", src, out, cyc);
    run<0, 0, 0, 12, 0, 0, 0>("MFMA + 12 pk_fma", src, out, cyc);
    run<0, 0, 0, 16, 0, 0, 0>("MFMA + 16 pk_fma", src, out, cyc);
    run<0, 0, 0,  8, 2, 0, 0>("MFMA + 8 pk_fma + 4 exp", src, out, cyc);
    run<4, 1, 1,  8, 1, 0, 0>("everything: 4 reads, write, load, 8 pk_fma, 3 exp", src, out, cyc);
    run<4, 1, 1,  8, 1, 1, 0>("everything (builtin MFMA)", src, out, cyc);
    run<4, 1, 1,  8, 1, 0, 0, 16>("everything, body unrolled x16 (128 groups of code)", src, out, cyc);
    run<4, 1, 1,  8, 1, 0, 0, 1, 1>("everything, random operands", src, out, cyc);
    run<4, 1, 1,  8, 1, 0, 0, 16, 1>("everything, x16 unrolled, random operands", src, out, cyc);
    run<0, 0, 0,  0, 0, 0, 0, 1, 1>("MFMA only, random operands", src, out, cyc);
    run<4, 0, 0,  0, 0, 0, 0, 1, 1>("MFMA + 4 reads, random operands", src, out, cyc);
    run<4, 0, 1,  0, 0, 0, 0, 1, 1>("MFMA + 4 reads + load, random operands", src, out, cyc);
    run<4, 1, 1,  0, 0, 0, 0, 1, 1>("MFMA + 4 reads + load + write, random operands", src, out, cyc);
    run<4, 1, 1, 12, 0, 0, 0, 1, 1>("MFMA + 4 reads + load + write + 12 pk_fma, random", src, out, cyc);
    run<4, 1, 1, 12, 1, 0, 0, 1, 1>("MFMA + 4 reads + load + write + 12 pk_fma + 3 exp, random", src, out, cyc);
    run<4, 1, 1, 12, 2, 0, 0, 1, 1>("MFMA + 4 reads + load + write + 12 pk_fma + 4 exp, random", src, out, cyc);
    run16<0, 0, 1>("16x16x32 MFMA only, ZERO operands", out, cyc);
    run16<0, 0, 0>("16x16x32 MFMA only, near-constant operands", out, cyc);
    run16<0, 1, 0>("16x16x32 MFMA only, random operands", out, cyc);
    run16<2, 1, 0>("16x16x32 MFMA + 2 fragment reads, random operands", out, cyc);
    run16<4, 1, 0>("16x16x32 MFMA + 4 reads (weights + activations), random", out, cyc);
    return 0;
}
