// Is the evaluators' weight-fragment stream bound by the L2 -> CU bandwidth or by the bytes a wave keeps in flight?
// Eight waves per CU (one workgroup, as "16s"), each reading its own 488 KB slice of a 3.9 MB stream cyclically in 4 KiB units
// (4 x global_load_dwordx4 per lane and unit) through a ring of D register stages (D - 1 units in flight per wave), consuming
// each unit with 16 v_mfma_f32_16x16x32_f16 (MFMA = 1: the k-loop of the single-pass tile without activations and epilogue)
// or with 4 v_xor (MFMA = 0: the stream alone).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o tools/probes/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned SLICE = 122 * 4096;      // bytes per wave: 122 units (the 512-wide net's 116 + padding)

template <int D, int J, bool MFMA>
__device__ __forceinline__ void step(half8 (&st)[D][4], const char *base, unsigned &off, f32x4 (&acc)[16], u32x4 &x, const half8 (&b)[4]) {
    const half8 *p = reinterpret_cast<const half8 *>(base + off);
#pragma unroll
    for (int i = 0; i < 4; ++i) st[(J + D - 1) % D][i] = p[64 * i];
    off += 4096;
    off = off == SLICE ? 0u : off;
    if (MFMA) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[4 * i + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(st[J][i], b[q], acc[4 * i + q], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) x ^= __builtin_bit_cast(u32x4, st[J][i]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int D, int J, bool MFMA>
struct Steps {
    static __device__ __forceinline__ void run(half8 (&st)[D][4], const char *base, unsigned &off, f32x4 (&acc)[16], u32x4 &x, const half8 (&b)[4]) {
        step<D, J, MFMA>(st, base, off, acc, x, b);
        if constexpr (J + 1 < D) Steps<D, J + 1, MFMA>::run(st, base, off, acc, x, b);
    }
};

template <int D, bool MFMA>
__global__ __launch_bounds__(512, 2) void k(const char *stream, float *out, int rounds) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char *base = stream + (size_t)wave * SLICE + lane * 16;
    half8 st[D][4];
    half8 b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)(0.21f * (float)(((lane * 11 + i * 17 + j * 31) * 2654435761u >> 20) & 255) / 64.f - 0.4f);
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 x = {0u, 0u, 0u, 0u};
    unsigned off = 0;
#pragma unroll
    for (int u = 0; u < D - 1; ++u) {
        const half8 *p = reinterpret_cast<const half8 *>(base + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) st[u][i] = p[64 * i];
        off += 4096;
    }
    for (int r = 0; r < rounds; ++r) Steps<D, 0, MFMA>::run(st, base, off, acc, x, b);
    float s = (float)(x[0] ^ x[1] ^ x[2] ^ x[3]);
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][2];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static char *g_stream;
static float *g_out;
template <int D, bool MFMA>
void run(int blocks) {
    const int units = 122 * 24;                 // per wave: 24 passes over its slice
    const int rounds = units / D;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<D, MFMA>), dim3(blocks), dim3(512), 0, 0, g_stream, g_out, rounds);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double bytes_cu = 8.0 * rounds * D * 4096.0;
    printf("%s  stages %2d (%2d KiB in flight per wave)  blocks %3d  %.3f ms  %6.1f GB/s per CU  %5.2f TB/s in all%s\n",
           MFMA ? "with 16 MFMAs per unit" : "stream alone          ", D, 4 * (D - 1), blocks, best, bytes_cu / best / 1e6,
           bytes_cu * blocks / best / 1e9, MFMA ? "" : "");
    if (MFMA) printf("        -> %.1f ns per unit and wave pair (16 MFMAs each: 512 matrix-pipe cycles per SIMD)\n", best * 1e6 / (rounds * D));
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    (void)hipMalloc(&g_stream, 8 * SLICE + 65536);
    (void)hipMalloc(&g_out, sizeof(float) * 512 * 256);
    std::vector<unsigned short> h((8 * SLICE + 65536) / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x2c00 + ((i * 2654435761u) >> 22 & 0x3ff) + ((i & 1) << 15));
    (void)hipMemcpy(g_stream, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<3, false>(blocks);
    run<4, false>(blocks);
    run<6, false>(blocks);
    run<8, false>(blocks);
    run<12, false>(blocks);
    run<3, true>(blocks);
    run<4, true>(blocks);
    run<6, true>(blocks);
    run<8, true>(blocks);
    run<11, true>(blocks);
    return 0;
}
