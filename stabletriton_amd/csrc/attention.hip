// Fused attention core for gfx950: out = softmax(q k^T * scale) v, per head, no
// mask, head_dim 64 (every SDXL-base/refiner head).  Row A of SURVEY.md 8a:
// replaces the eager matmul/softmax/matmul of unet_pt.py:133-142 without ever
// materialising the (B,H,T,S) score tensor.
//
// bf16 kernel (flash style, v_mfma_f32_32x32x16_bf16):
//   * one wave owns 32 query rows; a block of NW waves walks the keys in tiles
//     of 64 staged through LDS (register-staged double buffer, K and V images
//     XOR-swizzled per 16-byte chunk so the fragment reads are conflict-free);
//   * scores are computed transposed, S^T = K Q^T, so a lane holds 32 scores of
//     ONE query row: row max / row sum are in-register plus one half-wave swap;
//   * the S^T accumulator is reused directly as the B operand of O^T = V^T P^T
//     (no LDS round trip for P); V^T fragments come from ds_read_b64_tr_b16
//     on the row-major V image;
//   * keys beyond S (the 77-token text context) are zero-filled and masked.
// fp32 kernel ("strict" parity mode): plain FMA online softmax, one query row
// per thread.
#include "common.h"
#include <stdlib.h>

static inline int att_dev_env_int(const char* name, int dflt) {      // developer knobs: -DST_DEV_CONFIGS builds only
#ifdef ST_DEV_CONFIGS
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

#ifdef ST_PROBE
static unsigned long long* g_att_probe = nullptr;
extern "C" void st_debug_set_att_probe(void* p) { g_att_probe = (unsigned long long*)p; }
__device__ __forceinline__ unsigned long long att_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define AP_STAMP(v) unsigned long long v = att_now();
#define AP_ADD(a, t1, t0) a += (t1) - (t0);
#else
#define AP_STAMP(v)
#define AP_ADD(a, t1, t0)
#endif

static constexpr int ATT_D = 64;
static constexpr int ATT_KV = 64;          // keys per tile

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max / sum with the lane 32 (or 16) away, through v_permlane32_swap / v_permlane16_swap instead of a
// ds_bpermute: no LDS round trip and, above all, no s_waitcnt lgkmcnt(0) in the middle of the softmax
// (that wait also drains the V fragment reads still in flight).  After the swap of two copies of x a lane
// holds its own value in one register and its partner's in the other.  Inline asm: hipcc 7.2 folds the two
// results of the builtin into one value (the sum came out as 2x); s_nop 1 covers the VALU-write hazard.
__device__ __forceinline__ void att_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void att_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float xmax32(float x) { float a = x, b = x; att_swap32(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xmax16(float x) { float a = x, b = x; att_swap16(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xsum32(float x) { float a = x, b = x; att_swap32(a, b); return a + b; }
__device__ __forceinline__ float xsum16(float x) { float a = x, b = x; att_swap16(a, b); return a + b; }

typedef __attribute__((address_space(3))) const char lds_cchar;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lds_cchar*)p; }

// All sixteen transposed 4x16 block reads (ds_read_b64_tr_b16) of one 64-key V tile: two lane base
// addresses (d-block 0 / 1), the (k-step, key-half) position is an immediate offset.  Issued without
// a wait so the softmax VALU work runs under the LDS latency; v_tile_wait() retires them and pins
// every destination register behind the wait (cdna guide 5.7 form ii).
struct VTile { u32x2 r[4][2][2]; };      // [k-step s][d-block][key-half n]

__device__ __forceinline__ void v_tile_issue(VTile& v, unsigned base0, unsigned base1) {
#define TR(dst, base, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(base))
    TR(v.r[0][0][0], base0, 0);    TR(v.r[0][0][1], base0, 1024); TR(v.r[0][1][0], base1, 0);    TR(v.r[0][1][1], base1, 1024);
    TR(v.r[1][0][0], base0, 2048); TR(v.r[1][0][1], base0, 3072); TR(v.r[1][1][0], base1, 2048); TR(v.r[1][1][1], base1, 3072);
    TR(v.r[2][0][0], base0, 4096); TR(v.r[2][0][1], base0, 5120); TR(v.r[2][1][0], base1, 4096); TR(v.r[2][1][1], base1, 5120);
    TR(v.r[3][0][0], base0, 6144); TR(v.r[3][0][1], base0, 7168); TR(v.r[3][1][0], base1, 6144); TR(v.r[3][1][1], base1, 7168);
#undef TR
}

__device__ __forceinline__ void v_tile_wait(VTile& v) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v.r[0][0][0]), "+v"(v.r[0][0][1]), "+v"(v.r[0][1][0]), "+v"(v.r[0][1][1]),
                   "+v"(v.r[1][0][0]), "+v"(v.r[1][0][1]), "+v"(v.r[1][1][0]), "+v"(v.r[1][1][1]),
                   "+v"(v.r[2][0][0]), "+v"(v.r[2][0][1]), "+v"(v.r[2][1][0]), "+v"(v.r[2][1][1]),
                   "+v"(v.r[3][0][0]), "+v"(v.r[3][0][1]), "+v"(v.r[3][1][0]), "+v"(v.r[3][1][1])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ int swz_k(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_v(int row) { return ((row >> 1) & 1) << 2; }

typedef __attribute__((address_space(3))) void att_lds_void_t;
typedef __attribute__((address_space(1))) const void att_gbl_cvoid_t;
__device__ __attribute__((aligned(16))) unsigned int g_att_zero16[4] = {0u, 0u, 0u, 0u};

// Pipeline per 64-key tile t (one barrier per tile, three LDS tile buffers filled by LDS-DMA):
//   wait DMA(t+1) -> barrier -> issue DMA(t+2) -> read V(t) and K(t+1) fragments
//   -> S_next = K(t+1) Q^T (MFMA, asynchronous) -> softmax of S_cur on the VALU (runs under those MFMAs)
//   -> O^T += V(t)^T P^T -> S_cur = S_next.
// The buffer refilled in trip t held tile t-1, whose last reads (V fragments of trip t-1) are in
// registers before any wave reaches this trip's barrier; the DMA is issued after that barrier.
// KS = 2 splits the keys over two wave groups of NW waves each (same query rows, key tiles
// [0, n/2) and [n/2, n), each group with its own DMA ring); the groups merge their (O, m, l) through
// LDS at the end.  It doubles the waves of launches that would leave SIMDs with a single wave.
template <int NW, int KS = 1>
__global__ __launch_bounds__(NW * KS * 64) void attn_bf16_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                                 const bf16* __restrict__ V, bf16* __restrict__ O,
                                                                 int T, int S, long ldq, long ldk, long ldv, long ldo, float scale_log2e,
                                                                 unsigned long long* probe) {
    constexpr int TILE_B = ATT_KV * 128;                 // bytes of one K (or V) tile image
    constexpr int BUF_B = 2 * TILE_B;                    // K image + V image
    constexpr int PIECES = 16 / NW > 0 ? 16 / NW : 1;    // 1-KiB DMA pieces per wave per tile (8 K + 8 V pieces)
    static_assert(16 % NW == 0, "waves must divide the 16 DMA pieces of a tile");
    extern __shared__ __attribute__((aligned(16))) char lds_all[];      // KS rings of 3 tile buffers

    const int t_ = threadIdx.x, lane = t_ & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(t_ >> 6);
    const int kg = KS > 1 ? wave_all / NW : 0;           // key group of this wave
    const int wave = wave_all - kg * NW;
    char* lds = lds_all + kg * (3 * BUF_B);
    const int r32 = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int qrow = min(q0 + r32, T - 1);

    const bf16* Qb = Q + (size_t)b * T * ldq + (size_t)head * ATT_D;
    const bf16* Kb = K + (size_t)b * S * ldk + (size_t)head * ATT_D;
    const bf16* Vb = V + (size_t)b * S * ldv + (size_t)head * ATT_D;
    const bf16* zeros = reinterpret_cast<const bf16*>(g_att_zero16);

    // Q^T fragments: lane (q = r32, h) holds d = 16ks + 8h .. +7
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(Qb + (size_t)qrow * ldq + 16 * ks + 8 * h);

    const int nkt_all = (S + ATT_KV - 1) / ATT_KV;
    const int nkt = (nkt_all + KS - 1) / KS;              // trips of every key group (tiles past the end are all-masked)
    const int kt0 = kg * nkt;                              // first key tile of this group

    // ---- LDS-DMA of one tile: piece p = wave*PIECES + i; p < 8 -> K row block p, else V row block p-8.
    // The LDS image is lane-linear, so the chunk swizzle goes on the per-lane SOURCE address.
    const int lr = lane >> 3, pc = lane & 7;
    auto dma_tile = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int pce = wave * PIECES + i;
            const int isv = pce >> 3, rb = pce & 7;
            const int row = rb * 8 + lr;
            const int key = (kt0 + kt) * ATT_KV + row;
            const int c = pc ^ (isv ? swz_v(row) : swz_k(row));
            const bf16* src = isv ? Vb + (size_t)key * ldv + c * 8 : Kb + (size_t)key * ldk + c * 8;
            if (key >= S) src = zeros;
            __builtin_amdgcn_global_load_lds((att_gbl_cvoid_t*)src, (att_lds_void_t*)(lds + buf * BUF_B + isv * TILE_B + rb * 1024), 16, 0, 0);
        }
    };
    auto qk_tile = [&](int buf, f32x16& s0, f32x16& s1) {
        const char* kb = lds + buf * BUF_B;
        bf16x8 kf0[4], kf1[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = 2 * ks + h;
            const int ra = r32, rb_ = 32 + r32;
            kf0[ks] = *reinterpret_cast<const bf16x8*>(kb + ra * 128 + ((c ^ swz_k(ra)) << 4));
            kf1[ks] = *reinterpret_cast<const bf16x8*>(kb + rb_ * 128 + ((c ^ swz_k(rb_)) << 4));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[ks], qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[ks], qf[ks], s1, 0, 0, 0);
        }
    };

    f32x16 o0 = {0}, o1 = {0};
    float m = -1e30f, l = 0.f;
    dma_tile(0, 0);
    if (nkt > 1) dma_tile(1, 1);
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 s0 = {0}, s1 = {0};
    qk_tile(0, s0, s1);

#ifdef ST_PROBE
    unsigned long long pa = 0, pb_ = 0, pc_ = 0, pd = 0;
#endif
    int cur = 0;                                   // buffer of tile kt
    for (int kt = 0; kt < nkt; ++kt) {
        AP_STAMP(t0)
        const int nb = cur == 2 ? 0 : cur + 1;     // buffer of tile kt+1
        const int fb = nb == 2 ? 0 : nb + 1;       // buffer to refill with tile kt+2 (held tile kt-1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own pieces of tile kt+1 have landed
        __builtin_amdgcn_s_barrier();                              // ... and everyone's; tile kt-1 is dead
        if (kt + 2 < nkt) dma_tile(kt + 2, fb);
        // V^T fragments of this tile: sixteen transposed reads, retired after the softmax arithmetic
        const char* vb = lds + cur * BUF_B + TILE_B;
        VTile vt;
        {
            const int q4 = (lane & 15) >> 2;
            const int key = 4 * h + q4;
            const int ch0 = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
            const unsigned row = lds_addr(vb) + key * 128 + 8 * (lane & 1);
            v_tile_issue(vt, row + ((ch0 ^ swz_v(key)) << 4), row + (((ch0 + 4) ^ swz_v(key)) << 4));
        }
        // scores of the NEXT tile go to the matrix pipe now and run under this tile's softmax
        f32x16 n0 = {0}, n1 = {0};
        if (kt + 1 < nkt) qk_tile(nb, n0, n1);
        AP_STAMP(t1)
        // mask the tail keys (only the last tile can have any)
        if ((kt0 + kt + 1) * ATT_KV > S) {
            const int kbase = (kt0 + kt) * ATT_KV + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= S) s0[r] = -INFINITY;
                if (key + 32 >= S) s1[r] = -INFINITY;
            }
        }
        // ---- online softmax (scaled by scale*log2e, base-2 exponent) ----
        float mx = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
        mx = xmax32(mx);
        const float m_new = fmaxf(m, mx * scale_log2e);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = fast_exp2(fmaf(s0[r], scale_log2e, -m_new));
            s1[r] = fast_exp2(fmaf(s1[r], scale_log2e, -m_new));
            rs += s0[r] + s1[r];
        }
        // rescale only when some row's running max moved (exact: alpha == 1 otherwise)
        if (__any(m_new != m)) {
            const float alpha = fast_exp2(m - m_new);
            l *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            m = m_new;
        }
        l += rs;
        AP_STAMP(t2)
        // ---- O^T += V^T P^T ; k-step s covers keys 16s .. 16s+15 of the tile ----
        v_tile_wait(vt);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            bf16x8 pb;
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[j] = (bf16)((s_ < 2) ? s0[8 * (s_ & 1) + j] : s1[8 * (s_ & 1) + j]);
            u32x4 w0 = {vt.r[s_][0][0][0], vt.r[s_][0][0][1], vt.r[s_][0][1][0], vt.r[s_][0][1][1]};
            u32x4 w1 = {vt.r[s_][1][0][0], vt.r[s_][1][0][1], vt.r[s_][1][1][0], vt.r[s_][1][1][1]};
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), pb, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), pb, o1, 0, 0, 0);
        }
        AP_STAMP(t3)
        s0 = n0; s1 = n1;
        cur = nb;
        AP_ADD(pa, t1, t0) AP_ADD(pb_, t2, t1) AP_ADD(pc_, t3, t2)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ST_PROBE
    if (probe && lane == 0) {
        unsigned long long* o = probe + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW * 8 + wave * 8;
        o[0] = pa; o[1] = pb_; o[2] = pc_; o[3] = pd; o[4] = nkt;
    }
#endif

    if constexpr (KS > 1) {
        // merge the key groups: group 1 parks (O, m, l) in LDS (the rings are dead), group 0 folds them in
        __syncthreads();
        float* park = reinterpret_cast<float*>(lds_all) + (size_t)wave * (34 * 64) + lane;
        if (kg == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { park[r * 64] = o0[r]; park[(16 + r) * 64] = o1[r]; }
            park[32 * 64] = m; park[33 * 64] = l;
        }
        __syncthreads();
        if (kg == 1) return;
        const float m2 = park[32 * 64], l2 = park[33 * 64];
        const float m_new = fmaxf(m, m2);
        const float a1 = fast_exp2(m - m_new), a2 = fast_exp2(m2 - m_new);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] = o0[r] * a1 + park[r * 64] * a2;
            o1[r] = o1[r] * a1 + park[(16 + r) * 64] * a2;
        }
        l = l * a1 + l2 * a2;
    }
    l = xsum32(l);
    const float inv = 1.0f / l;
    if (q0 + r32 < T) {
        bf16* orow = O + (size_t)b * T * ldo + (size_t)(q0 + r32) * ldo + (size_t)head * ATT_D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 a_, c_;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a_[e] = (bf16)(o0[4 * g + e] * inv); c_[e] = (bf16)(o1[4 * g + e] * inv); }
            *reinterpret_cast<bf16x4*>(orow + 8 * g + 4 * h) = a_;
            *reinterpret_cast<bf16x4*>(orow + 32 + 8 * g + 4 * h) = c_;
        }
    }
}

// ---- 16-row variant (v_mfma_f32_16x16x32_bf16): one wave owns 16 query rows ----
// Half the rows per wave doubles the wave count of a launch (the SDXL shapes give only 1.25 waves per
// SIMD with 32-row waves) and halves the live accumulators (S 16 + O 16 registers), so several waves
// share a SIMD and one wave's MFMAs run under another's softmax without any software pipelining.
//   S^T[key][q] = K Q^T : A = K rows (lane: key = l&15, d = 32ks + 8g..), B = Q^T (lane: q = l&15, same d)
//                         D: lane (q = l&15, g = l>>4) holds keys 16kb + 4g + r
//   O^T[d][q]  += V^T P^T: B = P^T straight from the S registers of key blocks (2kp, 2kp+1): k-slot
//                         8g + j <-> key 32kp + 16(j>>2) + 4g + (j&3); A = V^T through two transposed
//                         4x16 block reads per fragment that follow the same key order.
__device__ __forceinline__ int swz_k16(int row) { return row & 7; }
__device__ __forceinline__ int swz_v16(int row) { return ((row >> 1) & 3) << 1; }

struct VTile16 { u32x2 r[2][4][2]; };      // [key pair-block kp][d-block][key-half n]

__device__ __forceinline__ void v16_issue(VTile16& v, unsigned b0, unsigned b1, unsigned b2, unsigned b3) {
#define TR(dst, base, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(base))
    TR(v.r[0][0][0], b0, 0);    TR(v.r[0][0][1], b0, 2048); TR(v.r[0][1][0], b1, 0);    TR(v.r[0][1][1], b1, 2048);
    TR(v.r[0][2][0], b2, 0);    TR(v.r[0][2][1], b2, 2048); TR(v.r[0][3][0], b3, 0);    TR(v.r[0][3][1], b3, 2048);
    TR(v.r[1][0][0], b0, 4096); TR(v.r[1][0][1], b0, 6144); TR(v.r[1][1][0], b1, 4096); TR(v.r[1][1][1], b1, 6144);
    TR(v.r[1][2][0], b2, 4096); TR(v.r[1][2][1], b2, 6144); TR(v.r[1][3][0], b3, 4096); TR(v.r[1][3][1], b3, 6144);
#undef TR
}

__device__ __forceinline__ void v16_wait(VTile16& v) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v.r[0][0][0]), "+v"(v.r[0][0][1]), "+v"(v.r[0][1][0]), "+v"(v.r[0][1][1]),
                   "+v"(v.r[0][2][0]), "+v"(v.r[0][2][1]), "+v"(v.r[0][3][0]), "+v"(v.r[0][3][1]),
                   "+v"(v.r[1][0][0]), "+v"(v.r[1][0][1]), "+v"(v.r[1][1][0]), "+v"(v.r[1][1][1]),
                   "+v"(v.r[1][2][0]), "+v"(v.r[1][2][1]), "+v"(v.r[1][3][0]), "+v"(v.r[1][3][1])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void attn16_bf16_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                              const bf16* __restrict__ V, bf16* __restrict__ O,
                                                              int T, int S, long ldq, long ldk, long ldv, long ldo, float scale_log2e) {
    constexpr int TILE_B = ATT_KV * 128;
    constexpr int BUF_B = 2 * TILE_B;
    constexpr int PIECES = 16 / NW;
    static_assert(NW <= 16 && 16 % NW == 0, "waves must divide the 16 DMA pieces of a tile");
    __shared__ __attribute__((aligned(16))) char lds[3 * BUF_B];

    const int t_ = threadIdx.x, lane = t_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t_ >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 16;
    const int qrow = min(q0 + c16, T - 1);

    const bf16* Qb = Q + (size_t)b * T * ldq + (size_t)head * ATT_D;
    const bf16* Kb = K + (size_t)b * S * ldk + (size_t)head * ATT_D;
    const bf16* Vb = V + (size_t)b * S * ldv + (size_t)head * ATT_D;
    const bf16* zeros = reinterpret_cast<const bf16*>(g_att_zero16);

    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(Qb + (size_t)qrow * ldq + 32 * ks + 8 * g);

    const int lr = lane >> 3, pc = lane & 7;
    auto dma_tile = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int pce = wave * PIECES + i;
            const int isv = pce >> 3, rb = pce & 7;
            const int row = rb * 8 + lr;
            const int key = kt * ATT_KV + row;
            const int c = pc ^ (isv ? swz_v16(row) : swz_k16(row));
            const bf16* src = isv ? Vb + (size_t)key * ldv + c * 8 : Kb + (size_t)key * ldk + c * 8;
            if (key >= S) src = zeros;
            __builtin_amdgcn_global_load_lds((att_gbl_cvoid_t*)src, (att_lds_void_t*)(lds + buf * BUF_B + isv * TILE_B + rb * 1024), 16, 0, 0);
        }
    };

    // lane-constant LDS offsets of the fragment reads
    const int k_off0 = c16 * 128 + (((0 + g) ^ swz_k16(c16)) << 4);       // d-step 0; key block kb adds kb*2048
    const int k_off1 = c16 * 128 + (((4 + g) ^ swz_k16(c16)) << 4);       // d-step 1
    const int vkey = 4 * g + (c16 >> 2);
    const int vsw = swz_v16(vkey);
    const int vrow = vkey * 128 + 8 * (c16 & 1);
    const int vbit = (c16 & 3) >> 1;
    const int v_off0 = vrow + (((0 ^ vsw) + vbit) << 4), v_off1 = vrow + (((2 ^ vsw) + vbit) << 4);
    const int v_off2 = vrow + (((4 ^ vsw) + vbit) << 4), v_off3 = vrow + (((6 ^ vsw) + vbit) << 4);

    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -1e30f, l = 0.f;
    const int nkt = (S + ATT_KV - 1) / ATT_KV;

    dma_tile(0, 0);
    if (nkt > 1) dma_tile(1, 1);

    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int nb = cur == 2 ? 0 : cur + 1;
        const int fb = nb == 2 ? 0 : nb + 1;
        // own pieces of tile kt have landed (tile kt+1 may still be in flight) ...
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // ... and everyone's; tile kt-1 is dead
        if (kt + 2 < nkt) dma_tile(kt + 2, fb);

        const char* kb_ = lds + cur * BUF_B;
        f32x4 s[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const bf16x8 ka = *reinterpret_cast<const bf16x8*>(kb_ + kb * 2048 + k_off0);
            const bf16x8 kc = *reinterpret_cast<const bf16x8*>(kb_ + kb * 2048 + k_off1);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[0], acc, 0, 0, 0);
            s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc, qf[1], acc, 0, 0, 0);
        }
        // V^T fragments: issued now, retired after the softmax arithmetic
        VTile16 vt;
        {
            const unsigned vb = lds_addr(lds + cur * BUF_B + TILE_B);
            v16_issue(vt, vb + v_off0, vb + v_off1, vb + v_off2, vb + v_off3);
        }
        if ((kt + 1) * ATT_KV > S) {
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
        mx = xmax32(xmax16(mx));
        const float m_new = fmaxf(m, mx * scale_log2e);
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[kb][r] = fast_exp2(fmaf(s[kb][r], scale_log2e, -m_new));
                rs += s[kb][r];
            }
        if (__any(m_new != m)) {
            const float alpha = fast_exp2(m - m_new);
            l *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[db][r] *= alpha;
            m = m_new;
        }
        l += rs;
        v16_wait(vt);
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            bf16x8 pb;
#pragma unroll
            for (int j = 0; j < 4; ++j) { pb[j] = (bf16)s[2 * kp][j]; pb[4 + j] = (bf16)s[2 * kp + 1][j]; }
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                u32x4 w = {vt.r[kp][db][0][0], vt.r[kp][db][0][1], vt.r[kp][db][1][0], vt.r[kp][db][1][1]};
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), pb, o[db], 0, 0, 0);
            }
        }
        cur = nb;
    }

    l = xsum32(xsum16(l));
    const float inv = 1.0f / l;
    if (q0 + c16 < T) {
        bf16* orow = O + (size_t)b * T * ldo + (size_t)(q0 + c16) * ldo + (size_t)head * ATT_D;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            bf16x4 a_;
#pragma unroll
            for (int e = 0; e < 4; ++e) a_[e] = (bf16)(o[db][e] * inv);
            *reinterpret_cast<bf16x4*>(orow + 16 * db + 4 * g) = a_;
        }
    }
}

// ---- fp32 strict kernel: thread = one query row, keys in tiles of 32 via LDS ----
__global__ __launch_bounds__(128) void attn_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                       const float* __restrict__ V, float* __restrict__ O, int T, int S,
                                                       long ldq, long ldk, long ldv, long ldo, float scale) {
    constexpr int KT = 32;
    __shared__ __attribute__((aligned(16))) float ks[KT][ATT_D];
    __shared__ __attribute__((aligned(16))) float vs[KT][ATT_D];
    const int head = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 128 + threadIdx.x;
    const int qrow = min(qi, T - 1);
    const float* qp = Q + (size_t)b * T * ldq + (size_t)qrow * ldq + (size_t)head * ATT_D;
    const float* Kb = K + (size_t)b * S * ldk + (size_t)head * ATT_D;
    const float* Vb = V + (size_t)b * S * ldv + (size_t)head * ATT_D;
    float q[ATT_D], o[ATT_D];
#pragma unroll
    for (int d = 0; d < ATT_D; d += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(qp + d);
        q[d] = v[0]; q[d + 1] = v[1]; q[d + 2] = v[2]; q[d + 3] = v[3];
        o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
    }
    float m = -1e30f, l = 0.f;
    for (int k0 = 0; k0 < S; k0 += KT) {
        __syncthreads();
        for (int i = threadIdx.x; i < KT * ATT_D / 4; i += 128) {
            const int row = i / (ATT_D / 4), c = (i - row * (ATT_D / 4)) * 4;
            const int key = k0 + row;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < S) {
                kv = *reinterpret_cast<const f32x4*>(Kb + (size_t)key * ldk + c);
                vv = *reinterpret_cast<const f32x4*>(Vb + (size_t)key * ldv + c);
            }
            *reinterpret_cast<f32x4*>(&ks[row][c]) = kv;
            *reinterpret_cast<f32x4*>(&vs[row][c]) = vv;
        }
        __syncthreads();
        const int nk = min(KT, S - k0);
        float sc[KT];
        float mx = -1e30f;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < ATT_D; ++d) a = fmaf(q[d], ks[j][d], a);
            a *= scale;
            sc[j] = j < nk ? a : -1e30f;
            mx = fmaxf(mx, sc[j]);
        }
        const float m_new = fmaxf(m, mx);
        const float alpha = expf(m - m_new);
        m = m_new;
        l *= alpha;
#pragma unroll
        for (int d = 0; d < ATT_D; ++d) o[d] *= alpha;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const float pj = j < nk ? expf(sc[j] - m_new) : 0.f;
            l += pj;
#pragma unroll
            for (int d = 0; d < ATT_D; ++d) o[d] = fmaf(pj, vs[j][d], o[d]);
        }
    }
    if (qi < T) {
        float* op = O + (size_t)b * T * ldo + (size_t)qi * ldo + (size_t)head * ATT_D;
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < ATT_D; d += 4) {
            f32x4 v = {o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
            *reinterpret_cast<f32x4*>(op + d) = v;
        }
    }
}

#ifdef ST_PROBE
#define ATT_PROBE_ARG g_att_probe
#else
#define ATT_PROBE_ARG nullptr
#endif

extern "C" int st_attention(const void* q, const void* k, const void* v, void* out, int B, int T, int S, int H, int D,
                            long ldq, long ldk, long ldv, long ldo, float scale, int dtype, void* stream) {
    ST_REQUIRE(q && k && v && out, "attention: null pointer");
    ST_REQUIRE(B > 0 && T > 0 && S > 0 && H > 0, "attention: bad shape B=%d T=%d S=%d H=%d", B, T, S, H);
    ST_REQUIRE(D == ATT_D, "attention: head_dim %d not supported (only %d)", D, ATT_D);
    ST_REQUIRE(H <= 65535 && B <= 65535, "attention: too many heads/batches for one launch");
    const int vec = dtype == ST_BF16 ? 8 : 4;
    ST_REQUIRE(ldq % vec == 0 && ldk % vec == 0 && ldv % vec == 0 && ldo % 4 == 0, "attention: strides must keep 16-byte alignment");
    ST_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0, "attention: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ST_BF16) {
        const float c = scale * 1.4426950408889634f;
        // waves (32 query rows each) per block: fewer rows per block = more blocks to spread over
        // the 256 CUs and to co-schedule (MFMA of one wave under the softmax VALU of another)
        static const int force_nw = att_dev_env_int("ST_ATT_NW", 0);
        // measured (tools/op_bench.py): the loop is latency-bound, so the K/V staging shared by more
        // waves wins over more blocks; 8 waves once that still leaves >= 128 blocks, else 4
        int nw = ((long)cdiv(T, 256) * H * B >= 128 && S > 256) ? 8 : 4;
        if (force_nw == 1 || force_nw == 2 || force_nw == 4 || force_nw == 8) nw = force_nw;
        static const int rows16_env = att_dev_env_int("ST_ATT_R16", -1);
        // measured (tools/op_bench.py): 16-row waves win for the 77-key text context (more waves for a
        // two-tile loop) and for the 4096-token level (2560 instead of 1280 waves)
        const int rows16 = rows16_env >= 0 ? rows16_env : ((S <= 256 || (T >= 4096 && S >= 4096)) ? 4 : 0);
        if (rows16 == 16) {
            hipLaunchKernelGGL(attn16_bf16_kernel<16>, dim3(cdiv(T, 256), H, B), dim3(1024), 0, st, (const bf16*)q, (const bf16*)k,
                               (const bf16*)v, (bf16*)out, T, S, ldq, ldk, ldv, ldo, c);
            return st_check_launch("attention");
        } else if (rows16 == 8) {
            hipLaunchKernelGGL(attn16_bf16_kernel<8>, dim3(cdiv(T, 128), H, B), dim3(512), 0, st, (const bf16*)q, (const bf16*)k,
                               (const bf16*)v, (bf16*)out, T, S, ldq, ldk, ldv, ldo, c);
            return st_check_launch("attention");
        } else if (rows16 == 4) {
            hipLaunchKernelGGL(attn16_bf16_kernel<4>, dim3(cdiv(T, 64), H, B), dim3(256), 0, st, (const bf16*)q, (const bf16*)k,
                               (const bf16*)v, (bf16*)out, T, S, ldq, ldk, ldv, ldo, c);
            return st_check_launch("attention");
        }
        constexpr size_t RING = 3 * 2 * ATT_KV * 128;       // one DMA ring: three (K, V) tile buffers
        // key split: launches that give most SIMDs a single wave (SDXL's 32x32 level: 160 blocks of 4 waves)
        // run two key groups per block instead - twice the waves, half the tiles each, one LDS merge
        static const int force_ks = att_dev_env_int("ST_ATT_KS", -1);
        const bool split = force_ks >= 0 ? force_ks == 2 : (nw == 4 && (long)cdiv(T, 128) * H * B <= 256 && S >= 512);
        if (split) {
            auto kfn = attn_bf16_kernel<4, 2>;
            static bool once = (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * RING)), true);
            (void)once;
            hipLaunchKernelGGL(kfn, dim3(cdiv(T, 128), H, B), dim3(512), 2 * RING, st, (const bf16*)q, (const bf16*)k,
                               (const bf16*)v, (bf16*)out, T, S, ldq, ldk, ldv, ldo, c, ATT_PROBE_ARG);
        } else if (nw == 8)
            hipLaunchKernelGGL(attn_bf16_kernel<8>, dim3(cdiv(T, 256), H, B), dim3(512), RING, st, (const bf16*)q, (const bf16*)k,
                               (const bf16*)v, (bf16*)out, T, S, ldq, ldk, ldv, ldo, c, ATT_PROBE_ARG);
        else if (nw == 4)
            hipLaunchKernelGGL(attn_bf16_kernel<4>, dim3(cdiv(T, 128), H, B), dim3(256), RING, st, (const bf16*)q, (const bf16*)k,
                               (const bf16*)v, (bf16*)out, T, S, ldq, ldk, ldv, ldo, c, ATT_PROBE_ARG);
        else if (nw == 2)
            hipLaunchKernelGGL(attn_bf16_kernel<2>, dim3(cdiv(T, 64), H, B), dim3(128), RING, st, (const bf16*)q, (const bf16*)k,
                               (const bf16*)v, (bf16*)out, T, S, ldq, ldk, ldv, ldo, c, ATT_PROBE_ARG);
        else
            hipLaunchKernelGGL(attn_bf16_kernel<1>, dim3(cdiv(T, 32), H, B), dim3(64), RING, st, (const bf16*)q, (const bf16*)k,
                               (const bf16*)v, (bf16*)out, T, S, ldq, ldk, ldv, ldo, c, ATT_PROBE_ARG);
    } else if (dtype == ST_F32) {
        hipLaunchKernelGGL(attn_f32_kernel, dim3(cdiv(T, 128), H, B), dim3(128), 0, st, (const float*)q, (const float*)k,
                           (const float*)v, (float*)out, T, S, ldq, ldk, ldv, ldo, scale);
    } else {
        return st_fail("attention: unsupported dtype %d", dtype);
    }
    return st_check_launch("attention");
}
