// Developer micro-benchmark: per-CU L2 -> LDS fill rate, LDS-DMA vs register staging.
// hipcc -O3 --offload-arch=gfx950 l2fill.hip -o l2fill && ./l2fill
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void fill(const char* __restrict__ src, size_t region, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // each block streams `iters` tiles of NW*4 KiB from a region shared by the blocks of its XCD group
    const size_t tile = (size_t)NW * 4096;
    size_t off = ((size_t)blockIdx.x * 7919 * tile) % region;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const char* g = src + off + (size_t)wave * 4096 + lane * 16;
        char* l = lds + ((it & 1) * NW + wave) * 4096;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(g + i * 1024), (lds_void_t*)(l + i * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            u32x4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const u32x4*>(g + i * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(l + i * 1024 + lane * 16) = r[i];
        }
        off += tile;
        if (off + tile > region) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc = *reinterpret_cast<unsigned*>(lds + threadIdx.x * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int NW>
void run(const char* name, const char* src, size_t region, int blocks, unsigned* sink) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    fill<MODE, NW><<<blocks, NW * 64, 2 * NW * 4096>>>(src, region, 10, sink);
    hipEventRecord(e0);
    fill<MODE, NW><<<blocks, NW * 64, 2 * NW * 4096>>>(src, region, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * iters * NW * 4096;
    printf("%-28s region %6.1f MB blocks %4d: %8.1f GB/s per block, %7.2f TB/s total\n", name, region / 1e6, blocks,
           bytes / blocks / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    char* buf; unsigned* sink;
    const size_t cap = 1ull << 30;
    hipMalloc(&buf, cap); hipMalloc(&sink, 4);
    hipMemset(buf, 1, cap);
    for (size_t region : {(size_t)2 << 20, (size_t)24 << 20, (size_t)512 << 20}) {
        for (int blocks : {256, 512}) {
            run<0, 8>("lds-dma  8 waves", buf, region, blocks, sink);
            run<1, 8>("reg-stage 8 waves", buf, region, blocks, sink);
            run<0, 4>("lds-dma  4 waves", buf, region, blocks, sink);
            run<1, 4>("reg-stage 4 waves", buf, region, blocks, sink);
        }
    }
    return 0;
}
