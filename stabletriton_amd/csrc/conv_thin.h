// GEMM-shaped operators: the direct conv for thin inputs (conv_in: Cin = 4).  Internal to csrc/.
#pragma once
#include "gemm_args.h"

// ---- direct conv for thin inputs (conv_in: Cin = 4, K = R*S*Cin = 36) ---------------------
// Weights sit in LDS as fp32 [K][Cout]; a thread owns one output pixel and a strip of 16 output
// channels at a time: its K input values stay in registers, weight reads are wave-wide broadcasts
// (all lanes of a wave work on the same channel strip).
template <typename T, int KMAX>
__global__ __launch_bounds__(256) void conv_thin_kernel(const GemmArgs p, int R, int chunk) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];       // [K][chunk]: this block's output channels
    const T* x = (const T*)p.A;
    const T* w = (const T*)p.W;
    const int K = p.K, N = p.N;
    const int n_lo = blockIdx.y * chunk, n_hi = min(N, n_lo + chunk);  // blockIdx.y splits the output channels
    for (int i = threadIdx.x; i < K * (n_hi - n_lo); i += 256) {
        const int nl = i / K, k = i - nl * K;
        wsm[k * chunk + nl] = Elem<T>::to_f(w[(size_t)(n_lo + nl) * K + k]);
    }
    __syncthreads();
    const int m = blockIdx.x * 256 + threadIdx.x;
    const bool live = m < p.M;
    const int mm = live ? m : 0;
    const int hw = p.Hout * p.Wout;
    const int img = mm / hw, rem = mm - img * hw;
    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
    float xin[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) xin[k] = 0.f;
    if (R == 3 && p.S == 3 && p.Cin == 4 && KMAX >= 36) {
        // the SDXL conv_in shape, fully unrolled: static register indices, one 4-channel load per tap
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                int iy = oy * p.stride - p.pad + r, ix = ox * p.stride - p.pad + s_;
                const int He = p.ups ? 2 * p.Hin : p.Hin, We = p.ups ? 2 * p.Win : p.Win;
                const bool ok = iy >= 0 && ix >= 0 && iy < He && ix < We;
                if (p.ups) { iy >>= 1; ix >>= 1; }
                const T* xp = x + (((size_t)img * p.Hin + (ok ? iy : 0)) * p.Win + (ok ? ix : 0)) * 4;
                float f[4];
                Out4<T>::load(xp, f);
#pragma unroll
                for (int c = 0; c < 4; ++c) xin[(r * 3 + s_) * 4 + c] = ok ? f[c] : 0.f;
            }
    } else {
        int k = 0;
        for (int r = 0; r < R; ++r)
            for (int s_ = 0; s_ < p.S; ++s_) {
                int iy = oy * p.stride - p.pad + r, ix = ox * p.stride - p.pad + s_;
                const int He = p.ups ? 2 * p.Hin : p.Hin, We = p.ups ? 2 * p.Win : p.Win;
                const bool ok = iy >= 0 && ix >= 0 && iy < He && ix < We;
                if (p.ups) { iy >>= 1; ix >>= 1; }
                const T* xp = x + (((size_t)img * p.Hin + (ok ? iy : 0)) * p.Win + (ok ? ix : 0)) * p.Cin;
                for (int c = 0; c < p.Cin; ++c, ++k) {
                    const float v = ok ? Elem<T>::to_f(xp[c]) : 0.f;
#pragma unroll
                    for (int kk = 0; kk < KMAX; ++kk) if (kk == k) xin[kk] = v;     // keep xin[] in registers
                }
            }
    }
    for (int n0 = n_lo; n0 < n_hi; n0 += 16) {
        float acc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                const float xv = xin[k];
                const float* wr = wsm + k * chunk + (n0 - n_lo);
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + e);
                    acc[e] += xv * w4[0]; acc[e + 1] += xv * w4[1]; acc[e + 2] += xv * w4[2]; acc[e + 3] += xv * w4[3];
                }
            }
        }
        if (!live) continue;
#pragma unroll
        for (int e0 = 0; e0 < 16; e0 += 4) {
            const int co = n0 + e0;
            if (co >= N) break;
            float v[4] = {acc[e0], acc[e0 + 1], acc[e0 + 2], acc[e0 + 3]};
            if (p.epi & ST_EPI_BIAS)
                for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f(((const T*)p.bias)[co + e]);
            if (p.epi & ST_EPI_SILU)
                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
            if (p.epi & ST_EPI_ROWBIAS)
                for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f(((const T*)p.rowbias)[(size_t)(m / p.rows_per_batch) * N + co + e]);
            if (p.epi & ST_EPI_RESIDUAL)
                for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f(((const T*)p.residual)[(size_t)m * p.ldr + co + e]);
            Out4<T>::store((T*)p.C + (size_t)m * p.ldc + co, v);
        }
    }
}

template <typename T>
static int conv_thin_launch(const GemmArgs& a, int R, hipStream_t st) {
    // split the output channels over blockIdx.y until the launch has a few blocks per CU
    const int bx = cdiv(a.M, 256);
    int ny = cdiv(1024, bx);
    if (ny > a.N / 16) ny = a.N / 16;
    if (ny < 1) ny = 1;
    const int chunk = cdiv(cdiv(a.N, ny), 16) * 16;
    ny = cdiv(a.N, chunk);
    const size_t lds = (size_t)a.K * chunk * sizeof(float);
    ST_REQUIRE(lds <= 64 * 1024, "conv2d(thin): weights do not fit LDS");
    hipLaunchKernelGGL((conv_thin_kernel<T, 64>), dim3(bx, ny), dim3(256), lds, st, a, R, chunk);
    return st_check_launch("conv2d(thin)");
}
