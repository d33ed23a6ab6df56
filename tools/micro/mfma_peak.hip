// Calibration kernel for the MFMA-busy counter: every SIMD of every CU issues back-to-back bf16 MFMAs for the whole
// launch (one wave per SIMD, four independent accumulators), so SQ_VALU_MFMA_BUSY_CYCLES over the launch's cycles must
// read ~100 %.  tools/pmc_ops.sh profiles it next to the real kernels; tools/pmc_ops.py uses its ratio as the unit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void mfma_peak(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    if (hipMalloc(&out, 4) != hipSuccess) return 1;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(mfma_peak, dim3(256), dim3(256), 0, 0, out, iters);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    printf("done: 3 launches x 256 blocks x 4 waves x %d x 4 MFMAs\n", iters);
    return 0;
}
