// Developer microbenchmark (round 5): what does a kernel boundary cost after a kernel that leaves X MB of freshly stored output, by
// the cache policy of the stores?  The eight-phase GEMM's blocks live 32.5 us (first entry -> last exit, s_memrealtime) in a launch
// that takes 39.4 us back to back (profiles/r05_gemm8p_stamps.txt): 7 us per launch are outside every block's life, and they grow
// with the bytes the launch stored (2.9 us at 2 MB, 8 us at 63 MB).  On a multi-XCD agent the end-of-kernel release writes every
// dirty L2 line back (8 L2s, not coherent with each other): stores that write through would move that traffic under the epilogue.
// Each block spins `spin_us` (the K loop), then stores `bytes` with 16-byte stores (the epilogue), waits for the acknowledgement and
// stamps; a reader kernel (the consumer) then loads everything once.
//   kinds: 0 plain | 1 nt | 2 sc1 | 3 sc0 sc1 | 4 sc0 sc1 nt
//   hipcc --offload-arch=gfx950 -O3 kernel_boundary.hip -o kernel_boundary && ./kernel_boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long realtime() { return __builtin_amdgcn_s_memrealtime(); }      // 100 MHz

template <int KIND>
__global__ void __launch_bounds__(256) writer(char* out, unsigned long long* stamps, int bytes_per_block, int spin_ticks) {
    const unsigned long long t0 = realtime();
    while (realtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(2);
    char* p = out + (size_t)blockIdx.x * bytes_per_block + threadIdx.x * 16;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int off = 0; off < bytes_per_block; off += 4096, p += 4096) {
        if constexpr (KIND == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
        if constexpr (KIND == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
        if constexpr (KIND == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        if constexpr (KIND == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
        if constexpr (KIND == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = realtime(); }
}
__global__ void __launch_bounds__(256) reader(const char* in, unsigned* sink, unsigned long long* stamps, int bytes_per_block) {
    const unsigned long long t0 = realtime();
    const char* p = in + (size_t)blockIdx.x * bytes_per_block + threadIdx.x * 16;
    unsigned acc = 0;
    for (int off = 0; off < bytes_per_block; off += 4096, p += 4096) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p);
        acc += v[0] ^ v[3];
    }
    if (acc == 0x12345u) sink[0] = acc;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = realtime(); }
}
static double span_us(unsigned long long* d, int blocks) {
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    for (int b = 0; b < blocks; ++b) { lo = std::min(lo, h[2 * b]); hi = std::max(hi, h[2 * b + 1]); }
    return (hi - lo) / 100.0;
}
template <int KIND>
static void run(const char* name, int blocks, int bytes, int spin_us, char* buf, unsigned long long* st, unsigned long long* st2, unsigned* sink, bool with_reader) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    auto once = [&] {
        hipLaunchKernelGGL(writer<KIND>, dim3(blocks), dim3(256), 0, 0, buf, st, bytes, spin_us * 100);
        if (with_reader) hipLaunchKernelGGL(reader, dim3(blocks), dim3(256), 0, 0, buf, sink, st2, bytes);
    };
    for (int i = 0; i < 5; ++i) once();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) once();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e3 / reps, sw = span_us(st, blocks), sr = with_reader ? span_us(st2, blocks) : 0.0;
    printf("%-12s %4d blocks x %7d B = %6.1f MB | per %s %7.2f us | writer's blocks %6.2f us%s | outside the blocks %6.2f us\n", name, blocks, bytes,
           blocks * (double)bytes / 1e6, with_reader ? "pair  " : "launch", per, sw, with_reader ? (", reader's " + std::to_string(sr).substr(0, 5) + " us").c_str() : "", per - sw - sr);
}
int main() {
    char* buf; unsigned long long *st, *st2; unsigned* sink;
    hipMalloc(&buf, 256u << 20); hipMalloc(&st, 1 << 20); hipMalloc(&st2, 1 << 20); hipMalloc(&sink, 64);
    hipMemset(buf, 0, 256u << 20);
    const int sizes[] = {0, 8192, 32768, 131072, 262144};
    for (int with_reader = 0; with_reader < 2; ++with_reader)
        for (int blocks : {256, 1024})
            for (int bytes : sizes) {
                if (blocks == 1024 && bytes > 65536 * 2) continue;
                if (with_reader && bytes == 0) continue;
                run<0>("plain", blocks, bytes, 10, buf, st, st2, sink, with_reader);
                run<1>("nt", blocks, bytes, 10, buf, st, st2, sink, with_reader);
                run<2>("sc1", blocks, bytes, 10, buf, st, st2, sink, with_reader);
                run<3>("sc0 sc1", blocks, bytes, 10, buf, st, st2, sink, with_reader);
                run<4>("sc0 sc1 nt", blocks, bytes, 10, buf, st, st2, sink, with_reader);
                printf("\n");
            }
    return 0;
}
