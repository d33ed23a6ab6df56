// st_split_f32: fp32 rows -> split rows (csrc/split.h), the stand-alone producer of the strict mode's matrix operands
// (weights once per model; activations whose producer cannot emit the split image itself).  HBM-bound: 4 B in, 4 B out.
#include "split.h"

// one thread = eight consecutive values of one row: two 16-byte loads, one 16-byte store of hi halves, one of lo halves
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, char* __restrict__ out, long rows, int K, long ldx) {
    const int per_row = K >> 3;
    const long n = rows * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / per_row;
        const int k = (int)(i - r * per_row) * 8;
        const float* src = x + r * ldx + k;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        f16x8 hi, lo;
        split8(v, hi, lo);
        char* dst = out + (size_t)r * K * 4 + split_off(k);
        *reinterpret_cast<f16x8*>(dst) = hi;
        *reinterpret_cast<f16x8*>(dst + 64) = lo;
    }
}

extern "C" int st_split_f32(const float* x, void* xs, long rows, int K, long ldx, void* stream) {
    ST_REQUIRE(x && xs, "split_f32: null pointer");
    ST_REQUIRE(rows > 0 && K > 0 && K % 32 == 0, "split_f32: rows=%ld K=%d (K must be a multiple of 32)", rows, K);
    ST_REQUIRE(ldx >= K && ldx % 4 == 0 && ((uintptr_t)x | (uintptr_t)xs) % 16 == 0, "split_f32: rows must keep 16-byte alignment");
    const long n = rows * (K / 8);
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, (char*)xs, rows, K, ldx);
    return st_check_launch("split_f32");
}
