// Developer microbenchmark: issue cost (cycles per wave-instruction, one wave on a SIMD) of the vector instructions the
// attention softmax is made of, alone and beside back-to-back MFMAs.   hipcc --offload-arch=gfx950 -O3 issue_cost.hip -o issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(unsigned long long* out, const float* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)a0; fb[j] = (__bf16)a1; }
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    unsigned ldsoff = (threadIdx.x & 63) * 16;
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {   // 64 v_exp_f32, independent
            REP8(asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (MODE == 1) {   // 64 v_max3_f32
            REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %1, %1, %2, %3\n\tv_max3_f32 %2, %2, %3, %4\n\tv_max3_f32 %3, %3, %4, %5\n\tv_max3_f32 %4, %4, %5, %6\n\tv_max3_f32 %5, %5, %6, %7\n\tv_max3_f32 %6, %6, %7, %0\n\tv_max3_f32 %7, %7, %0, %1"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (MODE == 2) {   // 64 v_cvt_pk_bf16_f32
            REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %5, %6\n\tv_cvt_pk_bf16_f32 %2, %6, %7\n\tv_cvt_pk_bf16_f32 %3, %7, %4\n\tv_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %5, %6\n\tv_cvt_pk_bf16_f32 %2, %6, %7\n\tv_cvt_pk_bf16_f32 %3, %7, %4"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
        } else if constexpr (MODE == 3) {   // 64 MFMA 32x32x16 alternating two accumulators
            REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
                 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
                 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
                 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);)
        } else if constexpr (MODE >= 4 && MODE <= 7) {   // 64 x {MFMA + n exp}; n = MODE-3 ... 1,2,3,4
            REP8(REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                      asm volatile("v_exp_f32 %0, %0" : "+v"(a0));
                      if (MODE >= 5) asm volatile("v_exp_f32 %0, %0" : "+v"(a1));
                      if (MODE >= 6) asm volatile("v_exp_f32 %0, %0" : "+v"(a2));
                      if (MODE >= 7) asm volatile("v_exp_f32 %0, %0" : "+v"(a3));))
        } else if constexpr (MODE == 8) {   // 64 x {MFMA + 2 exp + 1 cvt + 1 max3}
            REP8(REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1\n\tv_max3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(r0), "+v"(a3) : "v"(a4), "v"(a5));))
        } else if constexpr (MODE == 9) {   // 64 x ds_read_b64_tr_b16
            REP8(REP8(asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(*(unsigned long long*)&r0) : "v"(ldsoff));))
        } else if constexpr (MODE == 10) {  // 64 x ds_read_b128
            REP8(REP8(asm volatile("ds_read_b128 %0, %1" : "=v"(*(__uint128_t*)&r0) : "v"(ldsoff));))
        } else if constexpr (MODE == 11) {  // 64 x LDS-DMA issue (1 KiB each) from an L2-resident line set
            REP8(REP8(asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(in + (threadIdx.x & 63) * 4), "s"(0u) : "memory");))
        } else if constexpr (MODE == 12) {  // 64 x {MFMA + 2 exp + cvt + 2 tr reads}
            REP8(REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1\n\tds_read_b64_tr_b16 %3, %4\n\tds_read_b64_tr_b16 %3, %4 offset:1024"
                                   : "+v"(a0), "+v"(a1), "+v"(r0), "=v"(*(unsigned long long*)&r2) : "v"(ldsoff));))
        } else if constexpr (MODE == 14) {  // MFMA (accumulator in AGPRs, A/B in VGPRs) + 2 exp + cvt + max3
            REP8(REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(fa), "v"(fb));
                      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1\n\tv_max3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(r0), "+v"(a3) : "v"(a4), "v"(a5));))
        } else if constexpr (MODE == 15) {  // MFMA (accumulator and A/B in AGPRs) + 2 exp + cvt + max3
            REP8(REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "a"(fa), "a"(fb));
                      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1\n\tv_max3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(r0), "+v"(a3) : "v"(a4), "v"(a5));))
        } else if constexpr (MODE == 16) {  // MFMA (all VGPR, asm) + 2 exp + cvt + max3
            REP8(REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1\n\tv_max3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(r0), "+v"(a3) : "v"(a4), "v"(a5));))
        } else if constexpr (MODE == 17) {  // two alternating accumulators (AGPR) + 2 exp + cvt + max3 per MFMA
            REP8(REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(fa), "v"(fb));
                      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1\n\tv_max3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(r0), "+v"(a3) : "v"(a4), "v"(a5));
                      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(fa), "v"(fb));
                      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1\n\tv_max3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(r0), "+v"(a3) : "v"(a4), "v"(a5));))
        } else if constexpr (MODE == 18) {  // MFMA + 4 independent v_add_f32 (the guide's filler)
            REP8(REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(fa), "v"(fb));
                      asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));))
        } else if constexpr (MODE == 13) {  // 64 x {MFMA + 1 DMA}
            REP8(REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(in + (threadIdx.x & 63) * 4), "s"(0u) : "memory");))
        }
    }
    const unsigned long long t1 = now();
    float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[0] + (float)(r0 + r1 + r2 + r3);
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; }
    if (sink == 12345.678f) out[1] = 1;
}
template <int MODE>
void run(const char* what, int per_iter, unsigned long long* d, const float* in, int waves) {
    const int iters = 64;
    unsigned long long h[2];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 65536, 0, d, in, iters);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-52s waves/block %d: %7.1f cycles per instruction-group\n", what, waves, (double)h[0] / (iters * per_iter));
}
int main() {
    unsigned long long* d;
    float* in;
    hipMalloc(&d, 64);
    hipMalloc(&in, 1 << 20);
    hipMemset(in, 0, 1 << 20);
    for (int waves : {4, 8}) {     // 1: one wave on the CU; 4: one per SIMD; 8: two per SIMD
        run<0>("v_exp_f32", 64, d, in, waves);
        run<1>("v_max3_f32", 64, d, in, waves);
        run<2>("v_cvt_pk_bf16_f32", 64, d, in, waves);
        run<3>("v_mfma_f32_32x32x16_bf16", 64, d, in, waves);
        run<4>("MFMA + 1 exp", 64, d, in, waves);
        run<5>("MFMA + 2 exp", 64, d, in, waves);
        run<6>("MFMA + 3 exp", 64, d, in, waves);
        run<7>("MFMA + 4 exp", 64, d, in, waves);
        run<8>("MFMA + 2 exp + cvt_pk + max3", 64, d, in, waves);
        run<9>("ds_read_b64_tr_b16", 64, d, in, waves);
        run<10>("ds_read_b128", 64, d, in, waves);
        run<11>("global_load_lds_dwordx4 (m0 write + issue)", 64, d, in, waves);
        run<12>("MFMA + 2 exp + cvt_pk + 2 tr reads", 64, d, in, waves);
        run<13>("MFMA + 1 LDS-DMA", 64, d, in, waves);
        run<16>("asm MFMA all-VGPR + 2 exp + cvt + max3", 64, d, in, waves);
        run<14>("asm MFMA acc in AGPR + 2 exp + cvt + max3", 64, d, in, waves);
        run<15>("asm MFMA acc,A,B in AGPR + 2 exp + cvt + max3", 64, d, in, waves);
        run<17>("2 acc chains (AGPR), each MFMA + 2 exp + cvt + max3", 128, d, in, waves);
        run<18>("asm MFMA acc in AGPR + 4 v_add_f32", 64, d, in, waves);
    }
    return 0;
}
