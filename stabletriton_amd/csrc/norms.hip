// GroupNorm(+SiLU), LayerNorm and GEGLU for gfx950.  All three are HBM-bound:
// 16-byte loads per lane, fp32 statistics, wave64 shuffles + LDS for the
// cross-wave step.  (Rows G and E of SURVEY.md section 8a, LayerNorm is 8f-1.)
#include "common.h"
#include "split.h"
#include <type_traits>

// =============================================================================
// GroupNorm
// -----------------------------------------------------------------------------
// Three launches: partial statistics (enough blocks to fill 256 CUs even though
// bs=1 has only 32 groups), a tiny Chan-combine of the partials, and the apply
// pass.  Partials are (mean, M2, count) so the combine is numerically stable and
// the result does not depend on the order blocks finish (deterministic).
// =============================================================================

static constexpr int GN_THREADS = 1024;
static constexpr int GN_MAX_BLOCKS = 256;

struct GnGeom {
    int VC;      // 16-byte vectors per pixel row (NHWC)
    int RP;      // pixel rows one block covers per pass
    int NB;      // blocks per image
    int P;       // pixels per block
};

template <typename T>
static GnGeom gn_geom(int C, int HW) {
    GnGeom g;
    g.VC = C / Elem<T>::VEC;
    g.RP = GN_THREADS / g.VC;
    if (g.RP < 1) g.RP = 1;
    int nb = (HW + g.RP - 1) / g.RP;
    g.NB = nb < GN_MAX_BLOCKS ? nb : GN_MAX_BLOCKS;
    g.P = (HW + g.NB - 1) / g.NB;
    g.NB = (HW + g.P - 1) / g.P;
    return g;
}

// NHWC partial statistics.  grid (NB, N), block 1024.  Thread (rp, col) owns one
// 16-byte channel vector `col` of pixel rows rp, rp+RP, ...; per-channel sums
// are then reduced over rp through LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_nhwc(const T* __restrict__ x, float4* __restrict__ part,
                                                            int C, int HW, int G, int VC, int RP, int P) {
    constexpr int VEC = Elem<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [RP][2][C]
    const int n = blockIdx.y, b = blockIdx.x, NB = gridDim.x;
    const int t = threadIdx.x;
    const int rp = t / VC, col = t - rp * VC;
    const int p0 = b * P, p1 = min(p0 + P, HW);
    float s[VEC], ss[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s[i] = 0.f; ss[i] = 0.f; }
    if (rp < RP) {
        const T* base = x + ((size_t)n * HW) * C + (size_t)col * VEC;
        for (int p = p0 + rp; p < p1; p += RP) {
            Vec16<T> v = load16(base + (size_t)p * C);
#pragma unroll
            for (int i = 0; i < VEC; ++i) { float f = v.get(i); s[i] += f; ss[i] += f * f; }
        }
        float* row = smem + (size_t)rp * 2 * C;
#pragma unroll
        for (int i = 0; i < VEC; ++i) { row[col * VEC + i] = s[i]; row[C + col * VEC + i] = ss[i]; }
    }
    __syncthreads();
    // channel sums over rp, fixed order
    for (int c = t; c < 2 * C; c += GN_THREADS) {
        float a = 0.f;
        for (int r = 0; r < RP; ++r) a += smem[(size_t)r * 2 * C + c];
        smem[c] = a;               // row 0 becomes the total (each c touched by one thread)
    }
    __syncthreads();
    if (t < G) {
        const int cpg = C / G;
        float S = 0.f, SS = 0.f;
        for (int i = 0; i < cpg; ++i) { S += smem[t * cpg + i]; SS += smem[C + t * cpg + i]; }
        const float cnt = (float)(max(p1 - p0, 0)) * (float)cpg;
        float mean = cnt > 0.f ? S / cnt : 0.f;
        float m2 = cnt > 0.f ? fmaxf(SS - S * mean, 0.f) : 0.f;
        part[((size_t)n * NB + b) * G + t] = make_float4(mean, m2, cnt, 0.f);
    }
}

// NCHW partial statistics: a group is one contiguous run of cpg*HW elements.
// grid (NB, N*G), block 256; scalar loads (no alignment assumption).
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_nchw(const T* __restrict__ x, float4* __restrict__ part,
                                                     int C, int HW, int G, long chunk) {
    const int ng = blockIdx.y, b = blockIdx.x, NB = gridDim.x;
    const int n = ng / G, g = ng - n * G;
    const int cpg = C / G;
    const long total = (long)cpg * HW;
    const T* base = x + ((size_t)n * C + (size_t)g * cpg) * HW;
    const long i0 = (long)b * chunk, i1 = min(i0 + chunk, total);
    float s = 0.f, ss = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) { float f = Elem<T>::to_f(base[i]); s += f; ss += f * f; }
    __shared__ float red[2][4];
    s = wave_sum(s); ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float S = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        float SS = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        float cnt = (float)max(i1 - i0, 0L);
        float mean = cnt > 0.f ? S / cnt : 0.f;
        float m2 = cnt > 0.f ? fmaxf(SS - S * mean, 0.f) : 0.f;
        part[((size_t)n * NB + b) * G + g] = make_float4(mean, m2, cnt, 0.f);
    }
}

// Chan et al. combine of the per-block partials -> (mean, rstd).  One wave per
// (image, group): lanes fold their partials, then a fixed butterfly merges lanes.
__device__ __forceinline__ void chan_merge(float& mean, float& m2, float& cnt, float pm, float p2, float pc) {
    const float tot = cnt + pc;
    if (pc > 0.f) {
        const float d = pm - mean;
        const float w = pc / tot;
        mean += d * w;
        m2 += p2 + d * d * (cnt * w);
        cnt = tot;
    }
}

__global__ __launch_bounds__(256) void gn_finalize(const float4* __restrict__ part, float2* __restrict__ stats, int NB, int G, int NG, float eps) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= NG) return;
    const int n = i / G, g = i - n * G;
    float mean = 0.f, m2 = 0.f, cnt = 0.f;
    for (int b = lane; b < NB; b += 64) {
        const float4 p = part[((size_t)n * NB + b) * G + g];
        chan_merge(mean, m2, cnt, p.x, p.y, p.z);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float pm = __shfl_xor(mean, o, 64), p2 = __shfl_xor(m2, o, 64), pc = __shfl_xor(cnt, o, 64);
        // both partners must compute the same merged value: order the pair by lane
        const bool low = (lane & o) == 0;
        float am = low ? mean : pm, a2 = low ? m2 : p2, ac = low ? cnt : pc;
        const float bm = low ? pm : mean, b2 = low ? p2 : m2, bc = low ? pc : cnt;
        chan_merge(am, a2, ac, bm, b2, bc);
        mean = am; m2 = a2; cnt = ac;
    }
    if (lane == 0) stats[i] = make_float2(mean, rsqrtf(m2 / cnt + eps));
}

template <typename T, bool SILU>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_nhwc(const T* __restrict__ x, const T* __restrict__ gamma,
                                                            const T* __restrict__ beta, const float2* __restrict__ stats,
                                                            T* __restrict__ y, int C, int HW, int G, int VC, int RP, int P,
                                                            const T* __restrict__ x1 = nullptr, int C0 = 0, char* __restrict__ ys = nullptr) {
    // ys != nullptr (fp32 only, C % 32 == 0): also the split image of y, rows = pixels (st_arm_split_output)
    // x1 != nullptr: the input is the channel concatenation [x | x1] that was never materialised - channels [0, C0) of a pixel
    // from x (pixel stride C0), the rest from x1 (pixel stride C - C0); a thread's channel vector lies in one of them
    constexpr int VEC = Elem<T>::VEC;
    const int n = blockIdx.y, b = blockIdx.x;
    const int t = threadIdx.x;
    const int rp = t / VC, col = t - rp * VC;
    if (rp >= RP) return;
    const int p0 = b * P, p1 = min(p0 + P, HW);
    const int cpg = C / G;
    float mu[VEC], a[VEC], bt[VEC];
    Vec16<T> gv = load16(gamma + col * VEC), bv = load16(beta + col * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float2 st = stats[n * G + (col * VEC + i) / cpg];
        mu[i] = st.x; a[i] = st.y * gv.get(i); bt[i] = bv.get(i);
    }
    const size_t base = ((size_t)n * HW) * C + (size_t)col * VEC;
    const bool second = x1 != nullptr && col * VEC >= C0;
    const int Cs = x1 == nullptr ? C : (second ? C - C0 : C0);          // pixel stride of this thread's source
    const T* __restrict__ xs = (second ? x1 + ((size_t)n * HW) * Cs + (size_t)(col * VEC - C0) : x + ((size_t)n * HW) * Cs + (size_t)col * VEC);
    for (int p = p0 + rp; p < p1; p += RP) {
        Vec16<T> v = load16(xs + (size_t)p * Cs), o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float f = (v.get(i) - mu[i]) * a[i] + bt[i];
            if (SILU) f = silu_f(f);
            o.set(i, f);
        }
        store16(y + base + (size_t)p * C, o);
        if constexpr (std::is_same<T, float>::value) {
            if (ys) {
                const float v4[4] = {o.get(0), o.get(1), o.get(2), o.get(3)};
                split_store4(ys + ((size_t)n * HW + p) * C * 4, col * VEC, v4);
            }
        }
    }
}

template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_nchw(const T* __restrict__ x, const T* __restrict__ gamma,
                                                     const T* __restrict__ beta, const float2* __restrict__ stats,
                                                     T* __restrict__ y, int C, int HW, int G) {
    const int nc = blockIdx.y;                 // n*C + c
    const int n = nc / C, c = nc - n * C;
    const float2 st = stats[n * G + c / (C / G)];
    const float a = st.y * Elem<T>::to_f(gamma[c]), bt = Elem<T>::to_f(beta[c]);
    const size_t base = (size_t)nc * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float f = (Elem<T>::to_f(x[base + i]) - st.x) * a + bt;
        if (SILU) f = silu_f(f);
        y[base + i] = Elem<T>::from_f(f);
    }
}

static size_t gn_ws_bytes(int N, int G) {
    // partials for up to GN_MAX_BLOCKS blocks per image + final (mean, rstd)
    return (size_t)N * GN_MAX_BLOCKS * G * sizeof(float4) + (size_t)N * G * sizeof(float2);
}

template <typename T>
static int gn_launch(const void* x, const void* gamma, const void* beta, void* y, int N, int C, int HW, int G,
                     float eps, int silu, int layout, void* ws, hipStream_t st, char* ys = nullptr) {
    float4* part = (float4*)ws;
    float2* stats = (float2*)((char*)ws + (size_t)N * GN_MAX_BLOCKS * G * sizeof(float4));
    int NB;
    if (layout == ST_NHWC) {
        ST_REQUIRE(C % Elem<T>::VEC == 0, "group_norm NHWC: C=%d must be a multiple of %d", C, Elem<T>::VEC);
        GnGeom g = gn_geom<T>(C, HW);
        ST_REQUIRE(g.VC <= GN_THREADS, "group_norm NHWC: C=%d too wide", C);
        size_t lds = (size_t)g.RP * 2 * C * sizeof(float);
        ST_REQUIRE(lds <= 160 * 1024, "group_norm NHWC: LDS %zu too large", lds);
        NB = g.NB;
        hipLaunchKernelGGL(gn_stats_nhwc<T>, dim3(NB, N), dim3(GN_THREADS), lds, st, (const T*)x, part, C, HW, G, g.VC, g.RP, g.P);
        hipLaunchKernelGGL(gn_finalize, dim3(cdiv(N * G, 4)), dim3(256), 0, st, part, stats, NB, G, N * G, eps);
        if (silu)
            hipLaunchKernelGGL((gn_apply_nhwc<T, true>), dim3(NB, N), dim3(GN_THREADS), 0, st, (const T*)x, (const T*)gamma,
                               (const T*)beta, stats, (T*)y, C, HW, G, g.VC, g.RP, g.P, (const T*)nullptr, 0, ys);
        else
            hipLaunchKernelGGL((gn_apply_nhwc<T, false>), dim3(NB, N), dim3(GN_THREADS), 0, st, (const T*)x, (const T*)gamma,
                               (const T*)beta, stats, (T*)y, C, HW, G, g.VC, g.RP, g.P, (const T*)nullptr, 0, ys);
    } else {
        const long total = (long)(C / G) * HW;
        NB = (int)((total + 8191) / 8192);
        if (NB > GN_MAX_BLOCKS) NB = GN_MAX_BLOCKS;
        int want = cdiv(1024, N * G);            // keep ~1k blocks in flight at small N*G
        if (NB > want && want >= 1) NB = want > 1 ? want : 1;
        long chunk = (total + NB - 1) / NB;
        NB = (int)((total + chunk - 1) / chunk);
        hipLaunchKernelGGL(gn_stats_nchw<T>, dim3(NB, N * G), dim3(256), 0, st, (const T*)x, part, C, HW, G, chunk);
        hipLaunchKernelGGL(gn_finalize, dim3(cdiv(N * G, 4)), dim3(256), 0, st, part, stats, NB, G, N * G, eps);
        int bx = cdiv(HW, 256 * 4);
        if (bx < 1) bx = 1;
        if (silu)
            hipLaunchKernelGGL((gn_apply_nchw<T, true>), dim3(bx, N * C), dim3(256), 0, st, (const T*)x, (const T*)gamma,
                               (const T*)beta, stats, (T*)y, C, HW, G);
        else
            hipLaunchKernelGGL((gn_apply_nchw<T, false>), dim3(bx, N * C), dim3(256), 0, st, (const T*)x, (const T*)gamma,
                               (const T*)beta, stats, (T*)y, C, HW, G);
    }
    return st_check_launch("group_norm");
}

extern "C" size_t st_group_norm_workspace_bytes(int N, int C, int HW, int groups) {
    (void)C; (void)HW;
    return gn_ws_bytes(N, groups);
}

extern "C" int st_group_norm(const void* x, const void* gamma, const void* beta, void* y, int N, int C, int HW,
                             int groups, float eps, int silu, int layout, int dtype, void* workspace, void* stream) {
    ST_REQUIRE(x && gamma && beta && y && workspace, "group_norm: null pointer");
    ST_REQUIRE(N > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "group_norm: bad shape N=%d C=%d HW=%d G=%d", N, C, HW, groups);
    ST_REQUIRE(groups <= 1024 && N * C <= 65535 && N <= 65535, "group_norm: shape exceeds launch limits");
    ST_REQUIRE(layout == ST_NCHW || layout == ST_NHWC, "group_norm: bad layout %d", layout);
    hipStream_t st = (hipStream_t)stream;
    void* ys = nullptr;
    if (int e = st_take_split_arm("group_norm", (long)N * HW, C, dtype == ST_F32 && layout == ST_NHWC && C % 32 == 0, &ys)) return e;
    if (dtype == ST_BF16) return gn_launch<bf16>(x, gamma, beta, y, N, C, HW, groups, eps, silu, layout, workspace, st);
    if (dtype == ST_F16) return gn_launch<f16>(x, gamma, beta, y, N, C, HW, groups, eps, silu, layout, workspace, st);
    if (dtype == ST_F32) return gn_launch<float>(x, gamma, beta, y, N, C, HW, groups, eps, silu, layout, workspace, st, (char*)ys);
    return st_fail("group_norm: unsupported dtype %d", dtype);
}

// GroupNorm whose statistics come from the producer: the GEMM / conv that wrote x also left, per tile row of its
// launch and per channel, (sum, sum of squares) of the values it stored (st_linear / st_conv2d `col_stats`).  A channel
// concatenation (the decoder's skip connections, unet_pt.py:352-357) is two such sources side by side.  One wave per
// (image, group) adds the partials of its channels in double precision (fixed order), and the apply pass is the usual
// one: the statistics launch and its read of x are gone.
struct GnSource { const float2* part; int C; int tiles_per_image; };

__global__ __launch_bounds__(256) void gn_cols_finalize(GnSource s0, GnSource s1, float2* __restrict__ stats, int G, int NG,
                                                        int cpg, double count, float eps) {
    // one block per (image, group): its cpg channels are contiguous in a partial row, so consecutive threads read
    // consecutive channels of one tile row (coalesced), 256 / cpg tile rows at a time
    const int i = blockIdx.x, t_ = threadIdx.x;
    const int n = i / G, g = i - n * G;
    const int c_lo = g * cpg;
    double S = 0.0, Q = 0.0;
    auto sweep = [&](const GnSource& src, int lo, int hi) {        // channels [lo, hi) of this source
        const int w = hi - lo;
        if (w <= 0) return;
        const float2* base = src.part + (size_t)n * src.tiles_per_image * src.C + lo;
        const int total = w * src.tiles_per_image;
        for (int k = t_; k < total; k += 256) {
            const int t = k / w, c = k - t * w;
            const float2 v = base[(size_t)t * src.C + c];
            S += (double)v.x; Q += (double)v.y;
        }
    };
    const int c_hi = c_lo + cpg;
    sweep(s0, min(c_lo, s0.C), min(c_hi, s0.C));
    sweep(s1, max(c_lo, s0.C) - s0.C, max(c_hi, s0.C) - s0.C);
    __shared__ double red[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { S += __shfl_xor(S, o, 64); Q += __shfl_xor(Q, o, 64); }
    if ((t_ & 63) == 0) { red[0][t_ >> 6] = S; red[1][t_ >> 6] = Q; }
    __syncthreads();
    if (t_ == 0) {
        S = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        Q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const double mean = S / count;
        const double var = fmax(Q / count - mean * mean, 0.0);
        stats[i] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
    }
    (void)NG;
}

template <typename T>
static int gn_from_stats_launch(const void* x, const void* gamma, const void* beta, void* y, int N, int C, int HW, int G, float eps,
                                int silu, GnSource s0, GnSource s1, void* ws, hipStream_t st, const void* x1 = nullptr, int C0 = 0,
                                char* ys = nullptr) {
    ST_REQUIRE(C % Elem<T>::VEC == 0, "group_norm_from_stats: C=%d must be a multiple of %d", C, Elem<T>::VEC);
    GnGeom g = gn_geom<T>(C, HW);
    ST_REQUIRE(g.VC <= GN_THREADS, "group_norm_from_stats: C=%d too wide", C);
    float2* stats = (float2*)ws;
    hipLaunchKernelGGL(gn_cols_finalize, dim3(N * G), dim3(256), 0, st, s0, s1, stats, G, N * G, C / G,
                       (double)(C / G) * (double)HW, eps);
    if (silu)
        hipLaunchKernelGGL((gn_apply_nhwc<T, true>), dim3(g.NB, N), dim3(GN_THREADS), 0, st, (const T*)x, (const T*)gamma,
                           (const T*)beta, stats, (T*)y, C, HW, G, g.VC, g.RP, g.P, (const T*)x1, C0, ys);
    else
        hipLaunchKernelGGL((gn_apply_nhwc<T, false>), dim3(g.NB, N), dim3(GN_THREADS), 0, st, (const T*)x, (const T*)gamma,
                           (const T*)beta, stats, (T*)y, C, HW, G, g.VC, g.RP, g.P, (const T*)x1, C0, ys);
    return st_check_launch("group_norm_from_stats");
}

extern "C" int st_group_norm_from_stats(const void* x, const void* gamma, const void* beta, void* y, int N, int C, int HW,
                                        int groups, float eps, int silu, int dtype, const float* stats0, int C0, int rows0,
                                        const float* stats1, int C1, int rows1, void* workspace, void* stream) {
    ST_REQUIRE(x && gamma && beta && y && workspace && stats0, "group_norm_from_stats: null pointer");
    ST_REQUIRE(N > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "group_norm_from_stats: bad shape N=%d C=%d HW=%d G=%d", N, C, HW, groups);
    ST_REQUIRE(groups <= 1024 && N <= 65535, "group_norm_from_stats: shape exceeds launch limits");
    ST_REQUIRE(C0 > 0 && rows0 > 0 && HW % rows0 == 0, "group_norm_from_stats: source 0 has %d rows per partial for HW=%d", rows0, HW);
    ST_REQUIRE((stats1 == nullptr && C1 == 0 && C0 == C) || (stats1 && C1 > 0 && rows1 > 0 && HW % rows1 == 0 && C0 + C1 == C),
               "group_norm_from_stats: sources cover %d + %d channels, input has %d", C0, C1, C);
    GnSource s0 = {(const float2*)stats0, C0, HW / rows0};
    GnSource s1 = {(const float2*)stats1, C1, stats1 ? HW / rows1 : 0};
    hipStream_t st = (hipStream_t)stream;
    void* ys = nullptr;
    if (int e = st_take_split_arm("group_norm_from_stats", (long)N * HW, C, dtype == ST_F32 && C % 32 == 0, &ys)) return e;
    if (dtype == ST_BF16) return gn_from_stats_launch<bf16>(x, gamma, beta, y, N, C, HW, groups, eps, silu, s0, s1, workspace, st);
    if (dtype == ST_F16) return gn_from_stats_launch<f16>(x, gamma, beta, y, N, C, HW, groups, eps, silu, s0, s1, workspace, st);
    if (dtype == ST_F32) return gn_from_stats_launch<float>(x, gamma, beta, y, N, C, HW, groups, eps, silu, s0, s1, workspace, st, nullptr, 0, (char*)ys);
    return st_fail("group_norm_from_stats: unsupported dtype %d", dtype);
}

// The same for an input that is the channel concatenation [x0 | x1] of two NHWC tensors (C0 and C1 = C - C0 channels, each
// with the statistics of its producer): the concatenated tensor is never written (torch.cat of the decoder's skip
// connections, unet_pt.py:352-357).  Bit-identical to st_group_norm_from_stats on torch.cat([x0, x1], 1).
extern "C" int st_group_norm_from_stats_cat(const void* x0, const void* x1, const void* gamma, const void* beta, void* y, int N, int C, int HW,
                                            int groups, float eps, int silu, int dtype, const float* stats0, int C0, int rows0,
                                            const float* stats1, int C1, int rows1, void* workspace, void* stream) {
    ST_REQUIRE(x0 && x1 && gamma && beta && y && workspace && stats0 && stats1, "group_norm_from_stats_cat: null pointer");
    ST_REQUIRE(N > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "group_norm_from_stats_cat: bad shape N=%d C=%d HW=%d G=%d", N, C, HW, groups);
    ST_REQUIRE(groups <= 1024 && N <= 65535, "group_norm_from_stats_cat: shape exceeds launch limits");
    ST_REQUIRE(C0 > 0 && C1 > 0 && C0 + C1 == C && rows0 > 0 && rows1 > 0 && HW % rows0 == 0 && HW % rows1 == 0,
               "group_norm_from_stats_cat: sources cover %d + %d channels, input has %d", C0, C1, C);
    const int vec = dtype == ST_F32 ? 4 : 8;
    ST_REQUIRE(C0 % vec == 0 && C1 % vec == 0, "group_norm_from_stats_cat: channel counts (%d, %d) must be multiples of %d", C0, C1, vec);
    GnSource s0 = {(const float2*)stats0, C0, HW / rows0};
    GnSource s1 = {(const float2*)stats1, C1, HW / rows1};
    hipStream_t st = (hipStream_t)stream;
    void* ys = nullptr;
    if (int e = st_take_split_arm("group_norm_from_stats_cat", (long)N * HW, C, dtype == ST_F32 && C % 32 == 0, &ys)) return e;
    if (dtype == ST_BF16) return gn_from_stats_launch<bf16>(x0, gamma, beta, y, N, C, HW, groups, eps, silu, s0, s1, workspace, st, x1, C0);
    if (dtype == ST_F16) return gn_from_stats_launch<f16>(x0, gamma, beta, y, N, C, HW, groups, eps, silu, s0, s1, workspace, st, x1, C0);
    if (dtype == ST_F32) return gn_from_stats_launch<float>(x0, gamma, beta, y, N, C, HW, groups, eps, silu, s0, s1, workspace, st, x1, C0, (char*)ys);
    return st_fail("group_norm_from_stats_cat: unsupported dtype %d", dtype);
}

// =============================================================================
// LayerNorm: one wave per row, row held in registers, exact two-pass variance.
// =============================================================================
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                 const T* __restrict__ beta, T* __restrict__ y, int rows, int C, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int VC = C / VEC;
    const T* xr = x + (size_t)row * C;
    Vec16<T> v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int vc = lane + 64 * j;
        if (vc < VC) {
            v[j] = load16(xr + vc * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) s += v[j].get(i);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (lane + 64 * j < VC) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) { float d = v[j].get(i) - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    T* yr = y + (size_t)row * C;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int vc = lane + 64 * j;
        if (vc < VC) {
            Vec16<T> g = load16(gamma + vc * VEC), b = load16(beta + vc * VEC), o;
#pragma unroll
            for (int i = 0; i < VEC; ++i) o.set(i, (v[j].get(i) - mean) * rstd * g.get(i) + b.get(i));
            store16(yr + vc * VEC, o);
        }
    }
}

template <typename T>
static int ln_launch(const void* x, const void* g, const void* b, void* y, int rows, int C, float eps, hipStream_t st) {
    constexpr int VEC = Elem<T>::VEC;
    ST_REQUIRE(C % VEC == 0, "layer_norm: C=%d must be a multiple of %d", C, VEC);
    const int nv = cdiv(C / VEC, 64);
    ST_REQUIRE(nv <= 8, "layer_norm: C=%d too wide", C);
    dim3 grid(cdiv(rows, 4)), block(256);
#define LN_CASE(NV) case NV: hipLaunchKernelGGL((ln_kernel<T, NV>), grid, block, 0, st, (const T*)x, (const T*)g, (const T*)b, (T*)y, rows, C, eps); break;
    switch (nv) { LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8) }
#undef LN_CASE
    return st_check_launch("layer_norm");
}

extern "C" int st_layer_norm(const void* x, const void* gamma, const void* beta, void* y, int rows, int C, float eps,
                             int dtype, void* stream) {
    ST_REQUIRE(x && gamma && beta && y, "layer_norm: null pointer");
    ST_REQUIRE(rows > 0 && C > 0, "layer_norm: bad shape rows=%d C=%d", rows, C);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ST_BF16) return ln_launch<bf16>(x, gamma, beta, y, rows, C, eps, st);
    if (dtype == ST_F16) return ln_launch<f16>(x, gamma, beta, y, rows, C, eps, st);
    if (dtype == ST_F32) return ln_launch<float>(x, gamma, beta, y, rows, C, eps, st);
    return st_fail("layer_norm: unsupported dtype %d", dtype);
}

// =============================================================================
// GEGLU: out = state * gelu_erf(gate), strided rows so no .contiguous() copies.
// =============================================================================
template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const T* __restrict__ state, const T* __restrict__ gate, T* __restrict__ out,
                                                    long nvec, int FV, long lds_, long ldg, long ldo) {
    constexpr int VEC = Elem<T>::VEC;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const long r = i / FV;
        const int c = (int)(i - r * FV) * VEC;
        Vec16<T> a = load16(state + r * lds_ + c), g = load16(gate + r * ldg + c), o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.set(k, a.get(k) * gelu_for<T>(g.get(k)));
        store16(out + r * ldo + c, o);
    }
}

template <typename T>
static int geglu_launch(const void* s, const void* g, void* o, int rows, int F, long lds_, long ldg, long ldo, hipStream_t st) {
    constexpr int VEC = Elem<T>::VEC;
    ST_REQUIRE(F % VEC == 0 && lds_ % VEC == 0 && ldg % VEC == 0 && ldo % VEC == 0,
               "geglu: F and row strides must be multiples of %d", VEC);
    ST_REQUIRE(((uintptr_t)s | (uintptr_t)g | (uintptr_t)o) % 16 == 0, "geglu: pointers must be 16-byte aligned");
    const long nvec = (long)rows * (F / VEC);
    int grid = (int)((nvec + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(geglu_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)s, (const T*)g, (T*)o, nvec, F / VEC, lds_, ldg, ldo);
    return st_check_launch("geglu");
}

extern "C" int st_geglu(const void* state, const void* gate, void* out, int rows, int F, long ld_state, long ld_gate,
                        long ld_out, int dtype, void* stream) {
    ST_REQUIRE(state && gate && out, "geglu: null pointer");
    ST_REQUIRE(rows > 0 && F > 0, "geglu: bad shape rows=%d F=%d", rows, F);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ST_BF16) return geglu_launch<bf16>(state, gate, out, rows, F, ld_state, ld_gate, ld_out, st);
    if (dtype == ST_F16) return geglu_launch<f16>(state, gate, out, rows, F, ld_state, ld_gate, ld_out, st);
    if (dtype == ST_F32) return geglu_launch<float>(state, gate, out, rows, F, ld_state, ld_gate, ld_out, st);
    return st_fail("geglu: unsupported dtype %d", dtype);
}
