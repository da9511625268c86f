// nefii_probe.hip - measurement only (include/nefii_amd.h: nefii_mfma_sustained_probe): what the matrix cores of this device
// sustain on dense fp16 MFMAs with random operands and nothing else in the instruction stream.  The tracer's evaluators are
// priced against the 2.5 PFLOP/s spec peak (bench.py roofline.frac); MI355X clocks down under dense MFMA work on real data
// (tools/probes/slot_probe.hip, profiles/r04/slot_probe.txt), so the line also carries what a bare MFMA loop reaches here.
#include <hip/hip_runtime.h>

#include "../../include/nefii_amd.h"

namespace {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// CH accumulator chains: a group is 2 x CH MFMAs (CH = 4: 8 per group, each chain's accumulator comes round again after 4
// instructions = 64 cycles of issue; the loop measures 148 cycles per 8 instead of the 128 a free-running pipe would take -
// profiles/r04/slot_probe.txt - CH = 8 leaves every dependency 128 cycles of room)
template <int CH>
__global__ __launch_bounds__(256, 1) void mfma_probe_kernel(float *sink, int groups) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[CH];
    half8 a[2], b[CH];
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() {
        h = h * 1664525u + 1013904223u;
        return (_Float16)((float)((h >> 9) & 0xffff) * (2.f / 65536.f) - 1.f);
    };
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) a[i][j] = rnd();
    for (int i = 0; i < CH; ++i) {
        for (int j = 0; j < 8; ++j) b[i][j] = rnd();
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    }
    // (inline assembly: with the builtin, hipcc 7.2 rotates the accumulators of this loop through overlapping register
    // ranges - a[16:19] <- a[14:17], a[12:15] <- a[10:13], ... - and the false dependencies hold the loop at 31 cycles per
    // MFMA instead of 16-18: round 4's sustained figure, 1.2 PFLOP/s at an undisturbed 2.39 GHz, was THAT loop's issue rate,
    // not the part's power limit - profiles/r05/fp8_probe.txt)
    for (int g = 0; g < groups; ++g) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < CH; ++q)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a[i]), "v"(b[q]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last results land before they are read
    float s = 0.f;
    for (int q = 0; q < CH; ++q)
        for (int i = 0; i < 4; ++i) s += acc[q][i];
    if (s == 12345.678f) sink[lane] = s;        // keeps the loop alive; never true in practice
}
}  // namespace

extern "C" int nefii_mfma_sustained_probe_chains(int groups, int chains, float *h_ms, double *h_flops, void *stream) {
    if (groups <= 0 || !h_ms || !h_flops || (chains != 4 && chains != 8)) return NEFII_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    float *sink = nullptr;
    hipError_t e = hipMalloc(&sink, 256 * sizeof(float));
    if (e != hipSuccess) return (int)e;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto launch = [&](int g) {
        if (chains == 4)
            hipLaunchKernelGGL(mfma_probe_kernel<4>, dim3(256), dim3(256), 0, st, sink, g);
        else
            hipLaunchKernelGGL(mfma_probe_kernel<8>, dim3(256), dim3(256), 0, st, sink, g);
    };
    launch(groups / 4 + 1);      // clocks settle
    (void)hipEventRecord(e0, st);
    launch(groups);
    (void)hipEventRecord(e1, st);
    e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return (int)e;
    *h_ms = ms;
    *h_flops = 256.0 * 4.0 * (double)groups * (2.0 * chains) * 16384.0;       // workgroups x waves x groups x MFMAs x 2 x 16 x 16 x 32
    return 0;
}

extern "C" int nefii_mfma_sustained_probe(int groups, float *h_ms, double *h_flops, void *stream) {
    return nefii_mfma_sustained_probe_chains(groups, 4, h_ms, h_flops, stream);
}
