// Developer microbenchmark (round 5): what does one LDS-DMA wave-instruction (1 KiB: 64 lanes x 16 B) cost a CU, by addressing form
// and by the number of waves that issue at once?  The eight-phase GEMM's K loop reads as (LDS-DMA issues x ~25 cycles) + (fragment
// reads at 256 B/clk), not overlapped with each other (profiles/r05_gemm8p_stamps.txt): 64 DMAs per 256 x 256 x 64 K tile = 1,600 of
// its 2,470 cycles.  Forms:   0  global_load_lds_dwordx4 v[addr:addr+1], off          (per-lane 64-bit address: what the kernels issue)
//                             1  global_load_lds_dwordx4 v_off, s[base:base+1]        (scalar base + per-lane 32-bit offset)
//                             2  buffer_load_dwordx4 v_off, s[rsrc:rsrc+3], 0 offen lds
//   hipcc --offload-arch=gfx950 -O3 dma_forms.hip -o dma_forms && ./dma_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define REP8(x) x x x x x x x x
template <int FORM>
__global__ void k(unsigned long long* out, const char* src, int iters, int region_bytes) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every block streams its own region (L2-resident after the first pass), a wave walks it in 1-KiB pieces
    const char* base = src + (size_t)blockIdx.x * region_bytes;
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane(wave * 1024);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "memory");
    unsigned off = (unsigned)(wave * 8192 + lane * 16);            // each wave its own 8 KiB window, 8 pieces of 1 KiB
    const unsigned long long sb = (unsigned long long)base;
    u32x4 rsrc = {(unsigned)sb, (unsigned)(sb >> 32) & 0xffffu, (unsigned)region_bytes, 0x00020000u};
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]); rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    __syncthreads();
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
        if constexpr (FORM == 0) {
            const char* p = base + off;
            asm volatile("global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\tglobal_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072\n\t"
                         "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\tglobal_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072" ::"v"(p) : "memory");
        } else if constexpr (FORM == 1) {
            asm volatile("global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072\n\t"
                         "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(off), "s"(sb) : "memory");
        } else {
            asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:1024 lds\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:2048 lds\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:3072 lds\n\t"
                         "buffer_load_dwordx4 %0, %1, 0 offen lds\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:1024 lds\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:2048 lds\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:3072 lds" ::"v"(off), "s"(rsrc) : "memory");
        }
        if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // (bounded queue: the vmcnt counter ends at 63)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = now();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}
template <int FORM>
static void run(int nw, int blocks, int iters, unsigned long long* dout, const char* src, int region) {
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(nw * 64), 16 * 1024, 0, dout, src, iters, region);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(nw * 64), 16 * 1024, 0, dout, src, iters, region);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 16);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per;
    for (int b = 0; b < blocks; ++b) {
        unsigned long long mx = 0;
        for (int w = 0; w < nw; ++w) mx = std::max(mx, h[b * 16 + w]);
        per.push_back((double)mx / (8.0 * iters * nw));            // CU cycles per wave-instruction with nw waves issuing
    }
    std::sort(per.begin(), per.end());
    printf("form %d  %d waves/CU  %3d blocks: %.1f cycles per 1-KiB DMA per CU (median; min %.1f max %.1f) = %.0f B/clk/CU\n", FORM, nw, blocks,
           per[per.size() / 2], per.front(), per.back(), 1024.0 / per[per.size() / 2]);
}
int main() {
    const int region = 64 * 1024, blocks = 256;
    char* src; unsigned long long* dout;
    hipMalloc(&src, (size_t)region * blocks + 65536);
    hipMemset(src, 1, (size_t)region * blocks + 65536);
    hipMalloc(&dout, blocks * 16 * 8);
    for (int nb : {1, 256})
        for (int nw : {1, 4, 8}) {
            run<0>(nw, nb, 256, dout, src, region);
            run<1>(nw, nb, 256, dout, src, region);
            run<2>(nw, nb, 256, dout, src, region);
        }
    return 0;
}
