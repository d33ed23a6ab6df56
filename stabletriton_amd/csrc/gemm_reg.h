// GEMM-shaped operators: the register-staged kernel (ragged K, i.e. K not a multiple of the 128-byte K tile) and its launcher.
// Internal to csrc/.
#pragma once
#include "epilogue.h"

template <typename T, int BM, int BN, int WGM, int WGN, bool CONV, bool GEGLU>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_kernel(const GemmArgs p) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KB = 8 * VEC;                     // elements per 128-byte row segment
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_IT = (BM * 8 + NT - 1) / NT, B_IT = (BN * 8 + NT - 1) / NT;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static_assert(!GEGLU || (TN % 2 == 0), "GEGLU pairs value/gate n-tiles inside one wave");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile_n = blockIdx.x / tiles_m, tile_m = blockIdx.x - tile_n * tiles_m;
    const int m0 = tile_m * BM;
    constexpr int BNO = GEGLU ? BN / 2 : BN;        // output columns per block
    const int n0 = tile_n * BNO;

    const T* __restrict__ Ap = (const T*)p.A;
    const T* __restrict__ Wp = (const T*)p.W;

    // ---- per-thread staging slots: fixed (row, chunk) for the whole K loop ----
    const T* a_ptr[A_IT];      // dense: row base + chunk offset.  conv: image base + chunk offset
    int a_iy[A_IT], a_ix[A_IT];
    int a_lds[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int id = t + i * NT;
        const int row = id >> 3, c = id & 7;
        const int m = m0 + row;
        a_ok[i] = (id < BM * 8) && (m < p.M);
        a_lds[i] = row * 128 + ((c ^ (row & 7)) << 4);
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int mm = a_ok[i] ? m : 0;
            const int img = mm / hw, rem = mm - img * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_iy[i] = oy * p.stride - p.pad;
            a_ix[i] = ox * p.stride - p.pad;
            a_ptr[i] = Ap + (size_t)img * p.Hin * p.Win * p.Cin + c * VEC;
        } else {
            a_iy[i] = c * VEC;          // k offset of this chunk inside the K step
            a_ix[i] = 0;
            a_ptr[i] = Ap + (size_t)(a_ok[i] ? m : 0) * p.lda + c * VEC;
        }
    }
    const T* b_ptr[B_IT];
    int b_lds[B_IT], b_k[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int id = t + i * NT;
        const int row = id >> 3, c = id & 7;
        int wrow;                         // row of W feeding LDS row `row`
        bool ok = id < BN * 8;
        if (GEGLU) {
            const int w_ = row / WTN, local = row - w_ * WTN;
            const int half = local >= WTN / 2 ? 1 : 0;
            const int ncol = n0 + w_ * (WTN / 2) + (local - half * (WTN / 2));
            ok = ok && ncol < p.N;
            wrow = ncol + half * p.Ng;
        } else {
            wrow = n0 + row;
            ok = ok && wrow < p.N;
        }
        b_ok[i] = ok;
        b_k[i] = c * VEC;
        b_lds[i] = A_BYTES + row * 128 + ((c ^ (row & 7)) << 4);
        b_ptr[i] = Wp + (size_t)(ok ? wrow : 0) * p.K + c * VEC;
    }

    u32x4 a_reg[A_IT], b_reg[B_IT];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_tile = [&](int kt) {
        const int k0 = kt * KB;
        if (CONV) {
            const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
            const int r = tap / p.S, s = tap - r * p.S;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int iy = a_iy[i] + r, ix = a_ix[i] + s;
                bool ok;
                if (p.ups) {
                    ok = a_ok[i] && iy >= 0 && ix >= 0 && iy < 2 * p.Hin && ix < 2 * p.Win;
                    iy >>= 1; ix >>= 1;
                } else {
                    ok = a_ok[i] && iy >= 0 && ix >= 0 && iy < p.Hin && ix < p.Win;
                }
                const T* src = a_ptr[i] + ((size_t)(ok ? iy : 0) * p.Win + (ok ? ix : 0)) * p.Cin + c0;
                a_reg[i] = ok ? *reinterpret_cast<const u32x4*>(src) : zero4;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const bool ok = a_ok[i] && (k0 + a_iy[i] < p.K);
                a_reg[i] = ok ? *reinterpret_cast<const u32x4*>(a_ptr[i] + k0) : zero4;
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const bool ok = b_ok[i] && (k0 + b_k[i] < p.K);
            b_reg[i] = ok ? *reinterpret_cast<const u32x4*>(b_ptr[i] + k0) : zero4;
        }
    };
    auto store_tile = [&](int buf) {
        char* base = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (t + i * NT < BM * 8) *reinterpret_cast<u32x4*>(base + a_lds[i]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (t + i * NT < BN * 8) *reinterpret_cast<u32x4*>(base + b_lds[i]) = b_reg[i];
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, q = lane >> 4;
    const int nk = (p.K + KB - 1) / KB;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const char* sa = lds + cur * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = 4 * kk + q;
            Frag fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WTM + i * 16 + r16;
                fa[i] = *reinterpret_cast<const Frag*>(sa + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WTN + j * 16 + r16;
                fb[j] = *reinterpret_cast<const Frag*>(sb + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], fb[j], fa[i]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue<T, TM, TN, WTM, WTN, GEGLU>(p, acc, m0, n0, wm, wn, r16, q);
}

template <typename T, int BM, int BN, int WGM, int WGN, bool CONV>
static void launch_cfg(const GemmArgs& a, hipStream_t st) {
    const size_t lds = 2 * (size_t)(BM + BN) * 128;
    const int tiles_m = cdiv(a.M, BM);
    if constexpr (!CONV) {
        if (a.epi & ST_EPI_GEGLU) {
            hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WGM, WGN, CONV, true>), dim3(tiles_m * cdiv(a.N, BN / 2)),
                               dim3(WGM * WGN * 64), lds, st, a);
            return;
        }
    }
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WGM, WGN, CONV, false>), dim3(tiles_m * cdiv(a.N, BN)), dim3(WGM * WGN * 64), lds, st, a);
}
