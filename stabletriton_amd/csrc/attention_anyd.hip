// Attention for the head sizes SDXL does not use: out = softmax(q k^T * scale) v per head for head_dim 16, 32 and 128, in
// bf16, fp16 and fp32.  Row A of SURVEY.md 8a: the reference's operator takes head_dim in {16, 32, 64, 128}
// (kernels/attention_fa2.py:118-123); 64 - every head of SDXL-base and of the refiner - runs on the tuned kernels of
// attention.hip / attention_f32.hip, the other three on this one kernel, written for coverage and not for the last cycle.
//
// Structure: the 16-row flash kernel of attention_core.h with the head size as a template parameter.
//   * a wave owns 16 query rows, a block four waves; K / V tiles of 64 keys go global -> registers -> LDS (rows padded by 16
//     bytes instead of swizzled; the loads of tile t+1 fly under the matrix work of tile t), one LDS buffer, two barriers per tile;
//   * scores transposed (S^T = K Q^T, contraction over head_dim in blocks of 32: head_dim 16 meets zero operands in the
//     upper half), lazy reference maximum, P^T from registers as the B operand of O^T = V^T P^T, V^T fragments by
//     ds_read_b64_tr_b16, row sums from the matrix pipe (an extra "d block" of V^T that is 1 in its first row);
//   * fp32 tensors (the strict mode) multiply as split operands exactly as attention_f32.hip does: x ~ hi + lo * 2^-11 in two
//     IEEE halves, three v_mfma_f32_16x16x32_f16 per product, fp32 softmax; the output's split image is written when a consumer
//     armed it (st_arm_split_output).
#include "attention_core.h"
#include "split.h"
#include <type_traits>

namespace {

template <typename TI, int D, int NW>
__global__ __launch_bounds__(NW * 64) void attn_anyd_kernel(const TI* __restrict__ Q, const TI* __restrict__ K, const TI* __restrict__ V,
                                                            TI* __restrict__ O, int T, int S, long ldq, long ldk, long ldv, long ldo,
                                                            float scale_log2e, char* __restrict__ Os, int Cs) {
    constexpr bool SP = std::is_same<TI, float>::value;                  // fp32 tensors: split operands
    typedef typename std::conditional<SP, f16, TI>::type E;              // the matrix pipe's element type
    typedef typename V16<E>::x8 E8;
    constexpr int NT = NW * 64;
    constexpr int PITCH = D * 2 + 16;                                    // bytes per key row of one image (16-byte pad: rows leave the bank pattern)
    constexpr int PLANE = ATT_KV * PITCH;
    constexpr int NPL = SP ? 2 : 1;                                      // images per matrix: hi (, lo)
    constexpr int NKS = D <= 32 ? 1 : D / 32;                            // 32-wide contraction steps of K Q^T
    constexpr int NDB = D / 16;                                          // 16-row blocks of O^T
    constexpr int CH = D / 8;                                            // 8-value chunks per key row
    constexpr int PIECES = 2 * ATT_KV * CH;                              // (K or V, key row, chunk) pieces of a tile
    constexpr int TASKS = (PIECES + NT - 1) / NT;
    static_assert(D == 16 || D == 32 || D == 64 || D == 128, "head sizes of the reference's operator");
    extern __shared__ __attribute__((aligned(16))) char lds[];           // 2 * NPL * PLANE bytes: K hi (, K lo), V hi (, V lo)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int head = blockIdx.y, b = blockIdx.z;
    const int c16 = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * NW * 16 + wave * 16;
    const int qrow = min(q0 + c16, T - 1);
    const TI* Kb = K + (size_t)b * S * ldk + (size_t)head * D;
    const TI* Vb = V + (size_t)b * S * ldv + (size_t)head * D;
    const E zero = (E)0.0f;

    // ---- Q fragments: d = 32 ks + 8 g .. + 7 of this lane's query row, times scale * log2(e); zero beyond head_dim ----------
    E8 qh[NKS], ql[NKS];
    {
        const TI* qp = Q + (size_t)b * T * ldq + (size_t)qrow * ldq + (size_t)head * D;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = 32 * ks + 8 * g;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (d0 < D) {
                if constexpr (SP) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(qp + d0), c = *reinterpret_cast<const f32x4*>(qp + d0 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = a[j] * scale_log2e; v[4 + j] = c[j] * scale_log2e; }
                } else {
                    const E8 raw = *reinterpret_cast<const E8*>(qp + d0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (float)raw[j] * scale_log2e;
                }
            }
            if constexpr (SP) {
                split8(v, qh[ks], ql[ks]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { qh[ks][j] = (E)v[j]; ql[ks][j] = zero; }
            }
        }
    }
    E8 ones, zeros8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ones[j] = (E)(c16 == 0 ? 1.0f : 0.0f); zeros8[j] = zero; }

    // ---- tile staging: piece = (K or V, key row, chunk of 8 values) -------------------------------------------------------
    const int nkt = (S + ATT_KV - 1) / ATT_KV;
    f32x4 stg[TASKS][SP ? 2 : 1];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < TASKS; ++i) {
            const int id = t + i * NT;
            if (PIECES % NT != 0 && id >= PIECES) continue;
            const bool isv = id >= PIECES / 2;
            const int rc = isv ? id - PIECES / 2 : id;
            const int row = rc / CH, c = rc % CH;
            const int key = kt * ATT_KV + row;
            const TI* src = (isv ? Vb + (size_t)min(key, S - 1) * ldv : Kb + (size_t)min(key, S - 1) * ldk) + c * 8;
            stg[i][0] = *reinterpret_cast<const f32x4*>(src);
            if constexpr (SP) stg[i][1] = *reinterpret_cast<const f32x4*>(src + 4);
            if (key >= S) {                         // masked keys: finite zeros
                stg[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (SP) stg[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < TASKS; ++i) {
            const int id = t + i * NT;
            if (PIECES % NT != 0 && id >= PIECES) continue;
            const bool isv = id >= PIECES / 2;
            const int rc = isv ? id - PIECES / 2 : id;
            const int row = rc / CH, c = rc % CH;
            char* dst = lds + (isv ? NPL * PLANE : 0) + row * PITCH + c * 16;
            if constexpr (SP) {
                const float v[8] = {stg[i][0][0], stg[i][0][1], stg[i][0][2], stg[i][0][3], stg[i][1][0], stg[i][1][1], stg[i][1][2], stg[i][1][3]};
                E8 hi, lo;
                split8(v, hi, lo);
                *reinterpret_cast<E8*>(dst) = hi;
                *reinterpret_cast<E8*>(dst + PLANE) = lo;
            } else {
                *reinterpret_cast<f32x4*>(dst) = stg[i][0];
            }
        }
    };

    // fragment offsets inside an image.  K (A operand of S^T): key row c16 of a 16-key block, values 32 ks + 8 g .. + 7.
    // V^T (A operand of O^T) through transposed 4 x 16 block reads: this lane addresses key 4 g + (c16 >> 2) (+ 16 for the second
    // read), values 16 db + 4 (c16 & 3) .. + 3 - the key order of the P^T registers (attention_core.h).
    const int k_off = c16 * PITCH + g * 16;
    const int v_off = (4 * g + (c16 >> 2)) * PITCH + 8 * (c16 & 3);
    const char* kimg = lds;
    const char* vimg = lds + NPL * PLANE;

    f32x4 om[NDB + 1], oc[NDB + 1];               // O^T d blocks and the row-sum block: main / correction accumulators
#pragma unroll
    for (int i = 0; i <= NDB; ++i) { om[i] = f32x4{0.f, 0.f, 0.f, 0.f}; oc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float m_ref = 0.f;

    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);       // in flight under this tile's matrix work

        // ---- scores: s[kb][r] = key 16 kb + 4 g + r against this lane's query row, minus m_ref
        f32x4 s[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 mn = {-m_ref, -m_ref, -m_ref, -m_ref}, cr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bool live = 32 * ks + 8 * g < D;                  // (head_dim 16: the upper half of the contraction is zero)
                const char* p = kimg + kb * 16 * PITCH + k_off + ks * 64;
                const E8 kh = live ? *reinterpret_cast<const E8*>(p) : zeros8;
                mn = AttMma<E>::m16(kh, qh[ks], mn);
                if constexpr (SP) {
                    const E8 kl = live ? *reinterpret_cast<const E8*>(p + PLANE) : zeros8;
                    cr = AttMma<E>::m16(kh, ql[ks], cr);
                    cr = AttMma<E>::m16(kl, qh[ks], cr);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][r] = SP ? __builtin_fmaf(cr[r], 1.0f / ST_SPLIT_SCALE, mn[r]) : mn[r];
        }
        if ((kt + 1) * ATT_KV > S) {                // mask the tail keys (only the last tile has any)
            const int kbase = kt * ATT_KV + 4 * g;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kbase + 16 * kb + r >= S) s[kb][r] = -INFINITY;
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][r]);
        if (kt == 0 || __any(mx > ATT_LAG)) {
            const float rmx = xmax32(xmax16(mx));
            const float delta = ((kt == 0 || rmx > ATT_LAG) && rmx > -INFINITY) ? rmx : 0.f;
            const float alpha = kt == 0 ? 1.f : fast_exp2(-delta);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kb][r] -= delta;
#pragma unroll
            for (int db = 0; db <= NDB; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) { om[db][r] *= alpha; oc[db][r] *= alpha; }
            m_ref += delta;
        }
        // ---- P = 2^s; O^T += V^T P^T
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = fast_exp2(s[2 * kp + (j >> 2)][j & 3]);
            E8 ph, pl;
            if constexpr (SP) {
                split8(pv, ph, pl);
            } else {
                ph = pack8<E>(pv[0], pv[1], pv[2], pv[3], pv[4], pv[5], pv[6], pv[7]);
                pl = zeros8;
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const int off = v_off + kp * 32 * PITCH + db * 32;
                const E8 a = v_frag<E>(vimg, off, off + 16 * PITCH);
                om[db] = AttMma<E>::m16(a, ph, om[db]);
                if constexpr (SP) {
                    const E8 c = v_frag<E>(vimg + PLANE, off, off + 16 * PITCH);
                    oc[db] = AttMma<E>::m16(a, pl, oc[db]);
                    oc[db] = AttMma<E>::m16(c, ph, oc[db]);
                }
            }
            om[NDB] = AttMma<E>::m16(ones, ph, om[NDB]);
            if constexpr (SP) oc[NDB] = AttMma<E>::m16(ones, pl, oc[NDB]);
        }
        __syncthreads();                            // every wave is done with this tile before the next one is stored over it
    }

    // row sum: row 0 of the extra block lives in register 0 of the lanes with g == 0
    const float l = __shfl(__builtin_fmaf(oc[NDB][0], 1.0f / ST_SPLIT_SCALE, om[NDB][0]), c16, 64);
    const float inv = 1.0f / l;
    if (q0 + c16 < T) {
        TI* orow = O + (size_t)b * T * ldo + (size_t)(q0 + c16) * ldo + (size_t)head * D;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            float a_[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) a_[e] = __builtin_fmaf(oc[db][e], 1.0f / ST_SPLIT_SCALE, om[db][e]) * inv;
            if constexpr (SP) {
                *reinterpret_cast<f32x4*>(orow + 16 * db + 4 * g) = f32x4{a_[0], a_[1], a_[2], a_[3]};
                if (Os) split_store4(Os + ((size_t)b * T + q0 + c16) * Cs * 4, head * D + 16 * db + 4 * g, a_);
            } else {
                typename V16<E>::x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (E)a_[e];
                *reinterpret_cast<typename V16<E>::x4*>(orow + 16 * db + 4 * g) = o4;
            }
        }
    }
}

template <typename TI, int D>
int launch_d(const void* q, const void* k, const void* v, void* out, int B, int T, int S, int H, long ldq, long ldk, long ldv, long ldo,
             float c, void* out_split, hipStream_t st) {
    constexpr size_t LDS = (size_t)2 * (std::is_same<TI, float>::value ? 2 : 1) * ATT_KV * (D * 2 + 16);      // (68 KiB for fp32 at head_dim 128)
    static unsigned long long lds_ok = 0;                                     // bit mask over device ordinals
    ensure_dynamic_lds(attn_anyd_kernel<TI, D, 4>, LDS, &lds_ok);
    hipLaunchKernelGGL((attn_anyd_kernel<TI, D, 4>), dim3(cdiv(T, 64), H, B), dim3(256), LDS, st, (const TI*)q, (const TI*)k, (const TI*)v,
                       (TI*)out, T, S, ldq, ldk, ldv, ldo, c, (char*)out_split, H * D);
    return st_check_launch("attention");
}

template <typename TI>
int launch_t(const void* q, const void* k, const void* v, void* out, int B, int T, int S, int H, int D, long ldq, long ldk, long ldv,
             long ldo, float c, void* out_split, hipStream_t st) {
    if (D == 16) return launch_d<TI, 16>(q, k, v, out, B, T, S, H, ldq, ldk, ldv, ldo, c, out_split, st);
    if (D == 32) return launch_d<TI, 32>(q, k, v, out, B, T, S, H, ldq, ldk, ldv, ldo, c, out_split, st);
#ifdef ST_DEV_CONFIGS
    if (D == 64) return launch_d<TI, 64>(q, k, v, out, B, T, S, H, ldq, ldk, ldv, ldo, c, out_split, st);      // (ST_ATT_ANYD: the generic kernel beside the tuned ones)
#endif
    if (D == 128) return launch_d<TI, 128>(q, k, v, out, B, T, S, H, ldq, ldk, ldv, ldo, c, out_split, st);
    return st_fail("attention: head_dim %d not supported (16, 32, 64, 128)", D);
}

}      // namespace

// (entry: st_attention for head_dim != 64, attention.hip)
int attention_anyd_launch(int dtype, const void* q, const void* k, const void* v, void* out, int B, int T, int S, int H, int D,
                          long ldq, long ldk, long ldv, long ldo, float scale, void* out_split, hipStream_t st) {
    const float c = scale * 1.4426950408889634f;
    if (dtype == ST_BF16) return launch_t<bf16>(q, k, v, out, B, T, S, H, D, ldq, ldk, ldv, ldo, c, nullptr, st);
    if (dtype == ST_F16) return launch_t<f16>(q, k, v, out, B, T, S, H, D, ldq, ldk, ldv, ldo, c, nullptr, st);
    if (dtype == ST_F32) return launch_t<float>(q, k, v, out, B, T, S, H, D, ldq, ldk, ldv, ldo, c, out_split, st);
    return st_fail("attention: unsupported dtype %d", dtype);
}
