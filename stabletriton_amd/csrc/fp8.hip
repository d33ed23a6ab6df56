// Activation quantisation for the fp8 projection path: OCP e4m3 bytes plus one fp32 scale per row,
//   scale_m = max_k |x[m][k]| / 448,  xq[m][k] = e4m3(x[m][k] / scale_m)   (round to nearest even, nothing saturates).
// One wave per row, the row held in registers (like ln_kernel); the LayerNorm variant normalises first, so the
// LayerNorm -> Linear pairs of the transformer blocks (unet_pt.py:192-208) cost one pass instead of two.
#include "common.h"

static constexpr float FP8_MAX = ST_FP8_MAX;

template <typename T, int NV, bool LN>
__global__ __launch_bounds__(256) void quant_fp8_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ gamma,
                                                        const T* __restrict__ beta, unsigned char* __restrict__ xq,
                                                        float* __restrict__ row_scale, int rows, int C, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int VC = C / VEC;
    const T* xr = x + (size_t)row * ldx;
    float v[NV][VEC];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int vc = lane + 64 * j;
        if (vc < VC) {
            const Vec16<T> t = load16(xr + vc * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) { v[j][i] = t.get(i); s += v[j][i]; }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[j][i] = 0.f;
        }
    }
    if (LN) {
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (lane + 64 * j < VC) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) { const float d = v[j][i] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int vc = lane + 64 * j;
            if (vc < VC) {
                const Vec16<T> g = load16(gamma + vc * VEC), b = load16(beta + vc * VEC);
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[j][i] = (v[j][i] - mean) * rstd * g.get(i) + b.get(i);
            }
        }
    }
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int i = 0; i < VEC; ++i) amax = fmaxf(amax, fabsf(v[j][i]));
    amax = wave_max(amax);
    const float scale = fmaxf(amax, 1e-12f) / FP8_MAX;
    const float inv = 1.0f / scale;
    if (lane == 0) row_scale[row] = scale;
    unsigned char* qr = xq + (size_t)row * C;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int vc = lane + 64 * j;
        if (vc < VC) {
            unsigned int w[VEC / 4];
#pragma unroll
            for (int i = 0; i < VEC / 4; ++i) w[i] = pack4_fp8(v[j][4 * i] * inv, v[j][4 * i + 1] * inv, v[j][4 * i + 2] * inv, v[j][4 * i + 3] * inv);
            if constexpr (VEC == 8) *reinterpret_cast<u32x2*>(qr + vc * VEC) = u32x2{w[0], w[1]};
            else *reinterpret_cast<unsigned int*>(qr + vc * VEC) = w[0];
        }
    }
}

template <typename T, bool LN>
static int quant_launch(const void* x, long ldx, const void* g, const void* b, void* xq, float* rs, int rows, int C, float eps, hipStream_t st) {
    constexpr int VEC = Elem<T>::VEC;
    ST_REQUIRE(C % VEC == 0 && ldx % VEC == 0, "quantize_fp8: C and the row stride must be multiples of %d", VEC);
    const int nv = cdiv(C / VEC, 64);
    ST_REQUIRE(nv <= 12, "quantize_fp8: C=%d too wide", C);
    dim3 grid(cdiv(rows, 4)), block(256);
#define Q_CASE(NV) case NV: hipLaunchKernelGGL((quant_fp8_kernel<T, NV, LN>), grid, block, 0, st, (const T*)x, ldx, (const T*)g, (const T*)b, (unsigned char*)xq, rs, rows, C, eps); break;
    switch (nv) { Q_CASE(1) Q_CASE(2) Q_CASE(3) Q_CASE(4) Q_CASE(5) Q_CASE(6) Q_CASE(7) Q_CASE(8) Q_CASE(9) Q_CASE(10) Q_CASE(11) Q_CASE(12) }
#undef Q_CASE
    return st_check_launch("quantize_fp8");
}

extern "C" int st_quantize_fp8(const void* x, long ldx, void* xq, float* row_scale, int rows, int C, int dtype, void* stream) {
    ST_REQUIRE(x && xq && row_scale && rows > 0 && C > 0, "quantize_fp8: bad arguments");
    ST_REQUIRE(((uintptr_t)x | (uintptr_t)xq) % 16 == 0, "quantize_fp8: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ST_BF16) return quant_launch<bf16, false>(x, ldx, nullptr, nullptr, xq, row_scale, rows, C, 0.f, st);
    if (dtype == ST_F16) return quant_launch<f16, false>(x, ldx, nullptr, nullptr, xq, row_scale, rows, C, 0.f, st);
    if (dtype == ST_F32) return quant_launch<float, false>(x, ldx, nullptr, nullptr, xq, row_scale, rows, C, 0.f, st);
    return st_fail("quantize_fp8: unsupported dtype %d", dtype);
}

extern "C" int st_layer_norm_quantize_fp8(const void* x, const void* gamma, const void* beta, void* xq, float* row_scale,
                                          int rows, int C, float eps, int dtype, void* stream) {
    ST_REQUIRE(x && gamma && beta && xq && row_scale && rows > 0 && C > 0, "layer_norm_quantize_fp8: bad arguments");
    ST_REQUIRE(((uintptr_t)x | (uintptr_t)xq) % 16 == 0, "layer_norm_quantize_fp8: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ST_BF16) return quant_launch<bf16, true>(x, C, gamma, beta, xq, row_scale, rows, C, eps, st);
    if (dtype == ST_F16) return quant_launch<f16, true>(x, C, gamma, beta, xq, row_scale, rows, C, eps, st);
    if (dtype == ST_F32) return quant_launch<float, true>(x, C, gamma, beta, xq, row_scale, rows, C, eps, st);
    return st_fail("layer_norm_quantize_fp8: unsupported dtype %d", dtype);
}


// ---- delayed per-tensor scaling (the fp8 plan of the compiled graph: optimizers/plan_fp8.py) ------------------------
// A producer's epilogue quantises with the scale derived from the PREVIOUS step's max |value| (no pass over the tensor, no
// extra launch, no host round trip) and leaves this step's maximum in the tensor's partial slots.  Once per step, before
// the first launch, this kernel turns the partials of every tensor into the scale of the coming step and clears them:
//   scale = margin * amax / 448 (what one e4m3 unit is worth), inv_scale = 1 / scale;  a tensor that saw nothing keeps its scale.
__global__ __launch_bounds__(ST_FP8_AMAX_SLOTS) void fp8_update_scales_kernel(float* __restrict__ scale, float* __restrict__ inv_scale,
                                                                               unsigned int* __restrict__ parts, float margin) {
    const int i = blockIdx.x, t = threadIdx.x;
    unsigned int* mine = parts + (size_t)i * ST_FP8_AMAX_SLOTS;
    float a = __uint_as_float(mine[t]);
    mine[t] = 0u;
    a = wave_max(a);
    __shared__ float red[ST_FP8_AMAX_SLOTS / 64];
    if ((t & 63) == 0) red[t >> 6] = a;
    __syncthreads();
    if (t == 0) {
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < ST_FP8_AMAX_SLOTS / 64; ++w) m = fmaxf(m, red[w]);
        if (m > 0.f) {
            const float s = margin * m / ST_FP8_MAX;
            scale[i] = s;
            inv_scale[i] = 1.0f / s;
        }
    }
}

extern "C" int st_fp8_update_scales(float* scale, float* inv_scale, unsigned int* amax_parts, int n_tensors, float margin, void* stream) {
    ST_REQUIRE(scale && inv_scale && amax_parts && n_tensors > 0 && margin >= 1.0f, "fp8_update_scales: bad arguments");
    hipLaunchKernelGGL(fp8_update_scales_kernel, dim3(n_tensors), dim3(ST_FP8_AMAX_SLOTS), 0, (hipStream_t)stream, scale, inv_scale, amax_parts, margin);
    return st_check_launch("fp8_update_scales");
}
