// nefii_probe.hip - measurement only (include/nefii_amd.h: nefii_mfma_sustained_probe): what the matrix cores of this device
// sustain on dense fp16 MFMAs with random operands and nothing else in the instruction stream.  The tracer's evaluators are
// priced against the 2.5 PFLOP/s spec peak (bench.py roofline.frac); MI355X clocks down under dense MFMA work on real data
// (tools/probes/slot_probe.hip, profiles/r04/slot_probe.txt), so the line also carries what a bare MFMA loop reaches here.
#include <hip/hip_runtime.h>

#include "../../include/nefii_amd.h"

namespace {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void mfma_probe_kernel(float *sink, int groups) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    half8 a[2], b[4];
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() {
        h = h * 1664525u + 1013904223u;
        return (_Float16)((float)((h >> 9) & 0xffff) * (2.f / 65536.f) - 1.f);
    };
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) a[i][j] = rnd();
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 8; ++j) b[i][j] = rnd();
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    }
    for (int g = 0; g < groups; ++g) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[q], acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 4; ++i) s += acc[q][i];
    if (s == 12345.678f) sink[lane] = s;        // keeps the loop alive; never true in practice
}
}  // namespace

extern "C" int nefii_mfma_sustained_probe(int groups, float *h_ms, double *h_flops, void *stream) {
    if (groups <= 0 || !h_ms || !h_flops) return NEFII_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    float *sink = nullptr;
    hipError_t e = hipMalloc(&sink, 256 * sizeof(float));
    if (e != hipSuccess) return (int)e;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(256), dim3(256), 0, st, sink, groups / 4 + 1);      // clocks settle
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(256), dim3(256), 0, st, sink, groups);
    (void)hipEventRecord(e1, st);
    e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return (int)e;
    *h_ms = ms;
    *h_flops = 256.0 * 4.0 * (double)groups * 8.0 * 16384.0;       // workgroups x waves x groups x MFMAs x 2 x 16 x 16 x 32
    return 0;
}
