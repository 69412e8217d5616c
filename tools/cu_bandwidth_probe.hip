// Probe (gfx950): how many bytes per second can ONE CU pull from L2 / MALL / HBM, as a function of the wavefronts it runs and the loads each
// keeps in flight?  One workgroup per CU (grid = CUs used), every thread streams 16-byte loads (U independent ones per iteration) over its
// workgroup's private slice; three working-set sizes: 2 MB per XCD in total (L2 hits after the first pass), 128 MB (MALL), 2 GB (HBM).
// Also: the same stream through the LDS DMA (global_load_lds, 16 B per lane) instead of VGPR loads.
//   hipcc --offload-arch=gfx950 -O3 tools/cu_bandwidth_probe.hip -o /tmp/cu_bw && /tmp/cu_bw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const char* src, char* dst) {
    typedef __attribute__((address_space(3))) void lds_void;
    __builtin_amdgcn_global_load_lds(src, (lds_void*)(uintptr_t)dst, 16, 0, 0);
}

template <int U, bool DMA>
__global__ void k_stream(const char* base, size_t slice_bytes, int passes, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const char* mine = base + (size_t)blockIdx.x * slice_bytes;
    const size_t step = (size_t)blockDim.x * 16;                   // bytes per "row" of the workgroup
    const size_t rows = slice_bytes / step;
    u32x4 acc = {0, 0, 0, 0};
    for (int p = 0; p < passes; ++p) {
        for (size_t r = 0; r + U <= rows; r += U) {
            if constexpr (DMA) {
#pragma unroll
                for (int u = 0; u < U; ++u) dma16(mine + (r + u) * step + threadIdx.x * 16, lds + ((u * blockDim.x + (threadIdx.x & ~63u)) * 16));
            } else {
                u32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const u32x4*>(mine + (r + u) * step + threadIdx.x * 16);
#pragma unroll
                for (int u = 0; u < U; ++u) acc ^= v[u];
            }
        }
    }
    if constexpr (DMA) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc[0] = reinterpret_cast<uint32_t*>(lds)[threadIdx.x]; }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

template <int U, bool DMA>
double run(const char* buf, size_t slice, int blocks, int threads, int passes, uint32_t* sink) {
    const size_t ldsb = DMA ? (size_t)U * threads * 16 : 0;
    if (DMA) hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stream<U, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipLaunchKernelGGL((k_stream<U, DMA>), dim3(blocks), dim3(threads), ldsb, 0, buf, slice, 1, sink);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL((k_stream<U, DMA>), dim3(blocks), dim3(threads), ldsb, 0, buf, slice, passes, sink);
    hipDeviceSynchronize();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return (double)slice * blocks * passes / s / 1e9;            // GB/s aggregate
}

int main() {
    const size_t total = (size_t)2 << 30;
    char* buf; uint32_t* sink;
    hipMalloc(&buf, total); hipMalloc(&sink, 4);
    hipMemset(buf, 1, total);
    struct Case { const char* name; size_t slice; int passes; } cases[] = {
        {"L2-resident (64 KB per CU)", 64 << 10, 4000}, {"MALL-resident (512 KB per CU, 128 MB)", 512 << 10, 400}, {"HBM (8 MB per CU, 2 GB)", 8 << 20, 8}};
    for (const Case& c : cases) {
        printf("== %s\n", c.name);
        for (int blocks : {256, 64}) {
            for (int threads : {256, 512, 1024}) {
                const double a = run<4, false>(buf, c.slice, blocks, threads, c.passes, sink);
                const double b = run<8, false>(buf, c.slice, blocks, threads, c.passes, sink);
                const double d = run<16, false>(buf, c.slice, blocks, threads, c.passes, sink);
                const double e = threads <= 512 ? run<8, true>(buf, c.slice, blocks, threads, c.passes, sink) : 0.0;
                const double f = threads <= 256 ? run<16, true>(buf, c.slice, blocks, threads, c.passes, sink) : 0.0;
                printf("  %3d CUs x %4d threads: VGPR loads U=4 %7.1f  U=8 %7.1f  U=16 %7.1f GB/s   LDS DMA U=8 %7.1f  U=16 %7.1f GB/s   (per CU: %5.1f / %5.1f / %5.1f / %5.1f / %5.1f)\n",
                       blocks, threads, a, b, d, e, f, a / blocks, b / blocks, d / blocks, e / blocks, f / blocks);
            }
        }
    }
    return 0;
}
