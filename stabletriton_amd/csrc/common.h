// Shared device/host helpers for the gfx950 operator library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/stabletriton_amd.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16;                    // IEEE half: the reference's own compute type (load_sdxl_pipeline.py:17-28 hands optimize_model a .half() module)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- host-side error plumbing -------------------------------------------------
int st_fail(const char* fmt, ...);      // records message, returns non-zero
int st_check_launch(const char* what);  // hipGetLastError -> status
int st_take_split_arm(const char* who, long rows, int cols, bool can_emit, void** out);      // runtime.hip: the image armed by st_arm_split_output

#define ST_REQUIRE(cond, ...) do { if (!(cond)) return st_fail(__VA_ARGS__); } while (0)

// ---- element traits -----------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;                 // elements per 16-byte vector
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Elem<bf16> {
    static constexpr int VEC = 8;
    static __device__ __forceinline__ float to_f(bf16 v) { return (float)v; }
    static __device__ __forceinline__ bf16 from_f(float v) { return (bf16)v; }
};

template <> struct Elem<f16> {
    static constexpr int VEC = 8;
    static __device__ __forceinline__ float to_f(f16 v) { return (float)v; }
    static __device__ __forceinline__ f16 from_f(float v) { return (f16)v; }
};
// 8- and 4-element vectors of a 16-bit element type (MFMA operands, 8-byte stores)
template <typename T> struct V16;
template <> struct V16<bf16> { typedef bf16x8 x8; typedef bf16x4 x4; };
template <> struct V16<f16> { typedef f16x8 x8; typedef f16x4 x4; };

// 16-byte vector of T, unpacked to floats and back
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    f32x4 v;
    static constexpr int N = 4;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16> {
    bf16x8 v;
    static constexpr int N = 8;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

template <> struct Vec16<f16> {
    f16x8 v;
    static constexpr int N = 8;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (f16)x; }
};

template <typename T>
__device__ __forceinline__ Vec16<T> load16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T>
__device__ __forceinline__ void store16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// erf to fp32 accuracy without branches (round 5; libm's erff is two divergent ranges and a full expf - the strict mode's GEGLU
// epilogue spent 28,700 cycles per 256 x 160 tile in it, a fifth of the block's life).  Two fits (tools/fit_erf.py: least squares at
// Chebyshev nodes in float64, rounded):  |a| < 0.95: a * Ps(a^2), degree 6;  else sign(a) * (1 - exp(-t * Pl(t))), t = min(|a|, 4.2),
// degree 8 (erfc(4.2) < 2^-25: 1.0 beyond).  Max |error| 1.5e-7 = 2.7 ulp over the line; the GELU built on it deviates 6.8e-7 from the
// exact function on [-6, 6] where torch's own fp32 GELU deviates 1.3e-6 (the rounding of 0.5 x (1 + erf)).
__device__ __forceinline__ float erf_f32(float a) {
    const float t = fminf(fabsf(a), 4.2f), s = a * a;
    float ps = 8.20979694253765e-05f;
    ps = fmaf(ps, s, -0.0008104441803880036f);
    ps = fmaf(ps, s, 0.005197642371058464f);
    ps = fmaf(ps, s, -0.026858031749725342f);
    ps = fmaf(ps, s, 0.11283671110868454f);
    ps = fmaf(ps, s, -0.37612631916999817f);
    ps = fmaf(ps, s, 1.128379225730896f);
    float pl = -2.0026573110953905e-07f;
    pl = fmaf(pl, t, 3.0096502996457275e-06f);
    pl = fmaf(pl, t, -5.605676278719329e-07f);
    pl = fmaf(pl, t, -0.0003319129755254835f);
    pl = fmaf(pl, t, 0.0038219974376261234f);
    pl = fmaf(pl, t, -0.024281442165374756f);
    pl = fmaf(pl, t, 0.10693735629320145f);
    pl = fmaf(pl, t, 0.6346867084503174f);
    pl = fmaf(pl, t, 1.1287704706192017f);
    const float e = __builtin_amdgcn_exp2f(pl * t * -1.4426950408889634f);
    return t < 0.95f ? a * ps : copysignf(1.0f - e, a);
}
// exact-erf GELU, as torch.nn.functional.gelu default (reference kernels/geglu.py:24)
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_f32(x * 0.70710678118654752f)); }
// Same function for the bf16 kernels: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below
// the 2^-9 rounding of the bf16 result) -- one v_rcp, one v_exp and a 5-term Horner instead of the
// branchy libm erff, which made the GEGLU epilogue VALU-bound (40 % of the GEMM's cycles).
__device__ __forceinline__ float gelu_erf_fast_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    const float erf_abs = fmaf(-poly * t, e, 1.0f);              // erf(|x| / sqrt 2)
    const float half_x = 0.5f * x;
    return fmaf(fabsf(half_x), erf_abs, half_x);                // 0.5 x (1 + sign(x) erf(|x|/sqrt 2))
}
template <typename T> __device__ __forceinline__ float gelu_for(float x) {
    if constexpr (sizeof(T) == 4) return gelu_erf_f(x); else return gelu_erf_fast_f(x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- e4m3 copies of an operator's output for an fp8 consumer (delayed per-tensor scaling) -------------------------------
static constexpr float ST_FP8_MAX = 448.0f;                 // largest finite OCP e4m3 value
static constexpr int ST_FP8_AMAX_SLOTS = 256;               // max |value| partials per tensor (spread atomics, reduced once per step)
__device__ __forceinline__ unsigned int pack4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);      // bytes 0, 1
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);       // bytes 2, 3
    return (unsigned int)w;
}
__device__ __forceinline__ float clamp_fp8(float v) { return fminf(fmaxf(v, -ST_FP8_MAX), ST_FP8_MAX); }
// a wave's max |value| -> one of the tensor's partial slots (non-negative floats order like their bit patterns)
__device__ __forceinline__ void publish_amax(unsigned int* slots, float a, int wave_id) {
    a = wave_max(a);
    if ((threadIdx.x & 63) == 0 && a > 0.f)
        __hip_atomic_fetch_max(slots + (wave_id & (ST_FP8_AMAX_SLOTS - 1)), __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Kernels that need more than 64 KiB of dynamic LDS have that limit raised with hipFuncSetAttribute, which is PER DEVICE:
// `done` is the caller's bit mask over device ordinals (one static per kernel instantiation), so a process that drives
// several devices raises it on each of them, once.
template <typename K>
static inline void ensure_dynamic_lds(K kernel, size_t bytes, unsigned long long* done) {
    if (bytes <= 64 * 1024) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done, __ATOMIC_RELAXED) & bit) return;
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    __atomic_fetch_or(done, bit, __ATOMIC_RELAXED);
}

// dtype code of the C ABI -> element size / checks shared by the entry points
static inline bool st_dtype_is16(int dtype) { return dtype == ST_BF16 || dtype == ST_F16; }
static inline bool st_dtype_ok(int dtype) { return dtype == ST_F32 || st_dtype_is16(dtype); }
// the GEMM-shaped entry points also take ST_F32S (split fp32 matrix operands; everything the epilogue touches is fp32)
static inline bool st_dtype_ok_gemm(int dtype) { return st_dtype_ok(dtype) || dtype == ST_F32S; }
