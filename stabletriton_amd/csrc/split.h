// Split fp32 values (ST_F32S): the matrix operands of the strict mode.  Internal to csrc/.
//
// A value x is held as two IEEE halves, x ~ hi + lo * 2^-11 with hi = f16(x) and lo = f16((x - hi) * 2^11): hi carries 11
// significant bits, the residual x - hi is exact in fp32, at most half an ulp of hi, and scaled by 2^11 it sits in the
// middle of the half range again whatever the magnitude of x, so lo adds 11 more bits: 22 in all (fp32: 24).  Small values:
// below 2^-14 both halves are subnormal and the absolute error is <= 2^-36.  Large values: |x| > 65504 has no half and gives
// inf / NaN - loud, never a silently saturated product (the strict mode is a parity mode).
//
// Memory layout of a row of K values (K % 32 == 0): per group of 32 consecutive k one 128-byte segment, bytes [0, 64) the 32
// hi halves, bytes [64, 128) the 32 lo halves - 4 bytes per value like fp32, and a 16-byte chunk c < 4 of a segment holds the
// hi halves of k = 8c .. 8c+7, chunk c + 4 their lo halves (what one lane of v_mfma_f32_16x16x32_f16 takes).
#pragma once
#include "common.h"

static constexpr float ST_SPLIT_SCALE = 2048.0f;      // 2^11

// (x passes through an empty asm: an opaque, ROUNDED fp32 value.  Where x is itself a product just computed - the GEGLU
//  epilogue's value * gelu(gate) - hipcc otherwise folds `x - hi` into fma(value, gelu, -hi), the residual of the UNROUNDED
//  product, and the image a producer leaves would differ in the last bits of lo from st_split_f32 of the fp32 value it stored;
//  HIP's __fsub_rn is a plain subtraction and does not stop that.)
__device__ __forceinline__ void split_f32(float x, f16& hi, f16& lo) {
    asm volatile("" : "+v"(x));
    hi = (f16)x;
    lo = (f16)((x - (float)hi) * ST_SPLIT_SCALE);
}

// eight consecutive values -> their 16 bytes of hi halves and 16 bytes of lo halves
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f16 h, l;
        split_f32(x[i], h, l);
        hi[i] = h; lo[i] = l;
    }
}

// four consecutive values of a row (k % 4 == 0) -> 8 bytes of hi halves at split_off(k) of the row's image, 8 bytes of lo halves 64 further
__device__ __forceinline__ void split_store4(char* row_image, int k, const float (&v)[4]) {
    f16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) { f16 h, l; split_f32(v[i], h, l); hi[i] = h; lo[i] = l; }
    char* dst = row_image + (size_t)(k >> 5) * 128 + (size_t)(k & 31) * 2;
    *reinterpret_cast<f16x4*>(dst) = hi;
    *reinterpret_cast<f16x4*>(dst + 64) = lo;
}

// byte offset of value k of a row inside the row's split image: hi half; the lo half sits 64 bytes further
__device__ __forceinline__ size_t split_off(int k) { return (size_t)(k >> 5) * 128 + (size_t)(k & 31) * 2; }
