// Where does `buffer_load_dwordx4 ... offen lds` land for M0 values beyond 64 KiB on gfx950?  (sdf_tile_c.h keeps a 128-KiB ring.)
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/dma_probe.hip -o tools/probes/dma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 1) void probe(const unsigned *src, unsigned *found, unsigned target) {
    __shared__ unsigned lds[38 * 1024];          // 152 KiB
    for (int i = threadIdx.x; i < 38 * 1024; i += 256) lds[i] = 0u;
    __syncthreads();
    if (threadIdx.x < 64) {
        __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(src), 0, 4096, 0x00020000);
        unsigned voff = threadIdx.x * 16, tmp;
        unsigned base = (unsigned)reinterpret_cast<size_t>(lds) + target, zero = 0;
        asm volatile("s_add_u32 m0, %1, 0\n\ts_add_u32 %0, %2, 0\n\tbuffer_load_dwordx4 %3, %4, %0 offen lds\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(tmp) : "s"(base), "s"(zero), "v"(voff), "s"(srd) : "memory");
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 38 * 1024; i += 256)
        if (lds[i] == 0xABC00000u) found[0] = i * 4;        // where word 0 of the fragment landed
    if (threadIdx.x == 0) found[1] = (unsigned)reinterpret_cast<size_t>(lds);
    for (int i = threadIdx.x; i < 38 * 1024; i += 256)
        if (lds[i] == 0xABC000FFu) found[2] = i * 4;        // word 255 (last lane's last dword)
}
int main() {
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0xABC00000u + (i < 256 ? i : 0x10000 + i);
    unsigned *d, *f;
    hipMalloc(&d, 4096); hipMalloc(&f, 16);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    const unsigned targets[] = {0, 1024, 32768, 65536 - 1024, 65536, 65536 + 4096, 98304, 131072 - 1024, 131072, 150 * 1024};
    for (unsigned t : targets) {
        unsigned init[4] = {0xFFFFFFFFu, 0, 0xFFFFFFFFu, 0};
        hipMemcpy(f, init, 16, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, f, t);
        unsigned r[4];
        hipMemcpy(r, f, 16, hipMemcpyDeviceToHost);
        printf("target %6u (lds base 0x%x): word0 landed at %d, word255 at %d  %s\n", t, r[1], (int)r[0], (int)r[2],
               r[0] == t && r[2] == t + 1020 ? "ok" : "MISMATCH");
    }
    return 0;
}
