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

#ifdef ST_PROBE
#define ATT_PROBE_ARG g_att_probe
#else
#define ATT_PROBE_ARG nullptr
#endif

#ifndef ST_ATT_PRIO
#define ST_ATT_PRIO 1
#endif
static constexpr int ATT_D = 64;
static constexpr int ATT_KV = 64;          // keys per tile

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

typedef __attribute__((address_space(3))) void att_lds_void_t;
typedef __attribute__((address_space(1))) const void att_gbl_cvoid_t;

// LDS-DMA issue in assembly.  The builtin form makes hipcc 7.2 treat every later ds_read_b64_tr_b16 (an intrinsic it
// takes for a possible LDS store) as dependent on the DMA: it puts s_waitcnt vmcnt(0) in front of the first transposed
// read after each issue, i.e. the wave waits out the flight time of the tile it has just requested.  Issued from asm,
// the DMA is invisible to that pass and only the kernel's own counted waits apply.  lds_off: wave-uniform byte offset.
__device__ __forceinline__ void att_dma16(const void* src, unsigned lds_off) {
#ifdef ATT_X_BUILTIN
    __builtin_amdgcn_global_load_lds((att_gbl_cvoid_t*)src, (att_lds_void_t*)(size_t)lds_off, 16, 0, 0);
    return;
#endif
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ unsigned att_lds_offset(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

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

// ---- 16-row kernel, second generation: the softmax is cut to what the VALU cannot avoid ------------
// At D = 64 the kernel is bound by the softmax arithmetic, not by the matrix pipe (per 16 x 64 score tile a wave
// issues 16-18 MFMAs = 290 pipe cycles, and the classic online softmax ~100 VALU instructions = 450 issue cycles).
// What is left here per score: one v_exp_f32, half a v_max3_f32, half a v_cvt_pk_bf16_f32:
//   * Q is pre-multiplied by scale * log2(e) once (bf16, like every MFMA operand), so scores are base-2 exponents;
//   * the S accumulators start at -m_ref (the row's reference maximum) instead of 0: the MFMA chain delivers
//     s - m_ref and the exponent needs no subtraction;
//   * m_ref follows the true row maximum lazily: a tile whose scores stay below m_ref + 2^ATT_LAG keeps it (softmax is
//     shift invariant; P <= 2^ATT_LAG is as exact in bf16 / fp32 as P <= 1); the first tile, and any tile that
//     exceeds the lag, takes the exact path (row maximum across lanes, rescale O, shift the pending scores);
//   * the row sums come out of the matrix pipe: a fifth "d block" of V^T that is 1 in its first row adds
//     sum_k P[k][q] to an accumulator (two MFMAs per tile instead of sixteen VALU adds), and being an accumulator
//     like O it is rescaled with O;
//   * the next tile's K Q^T is issued before this tile's softmax (runs under it).
// K/V staging (LDS-DMA ring of three tiles, one barrier per tile) is that of attn16_bf16_kernel.
// TAG only gives the cross-attention instantiation its own kernel name (profiles split the two).
static constexpr float ATT_LAG = 6.0f;

// V^T fragments through the compiler's own transposed LDS read (it places the two 8-byte halves of an MFMA operand
// in adjacent registers and counts the reads itself; the inline-asm form needed a v_mov per half)
typedef __attribute__((address_space(3))) bf16x4 att_lds_bf16x4;
__device__ __forceinline__ bf16x8 v_frag(const char* lds_base, int off_lo, int off_hi) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((att_lds_bf16x4*)(lds_base + off_lo));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((att_lds_bf16x4*)(lds_base + off_hi));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// eight probabilities -> one MFMA operand: four v_cvt_pk_bf16_f32
__device__ __forceinline__ bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    bf16x8 r;
    r[0] = (bf16)a0; r[1] = (bf16)a1; r[2] = (bf16)a2; r[3] = (bf16)a3; r[4] = (bf16)a4; r[5] = (bf16)a5; r[6] = (bf16)a6; r[7] = (bf16)a7;
    return r;
}

// Three-way maximum, deliberately NOT inline asm: the scores it reads come straight out of MFMAs, and the hardware does
// not interlock an MFMA result against a VALU read - the compiler inserts the wait states, but only for instructions it
// can see.  An asm v_max3_f32 here read accumulators that were still being written whenever the matrix pipe was shared
// with another kernel (tools/att_race.py: output changed by 1 ulp when a second stream kept the CUs busy).
__device__ __forceinline__ float att_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <int NW, int TAG>
__global__ __launch_bounds__(NW * 64) void attn16v2_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
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
    for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 raw = *reinterpret_cast<const bf16x8*>(Qb + (size_t)qrow * ldq + 32 * ks + 8 * g);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (bf16)((float)raw[j] * scale_log2e);
    }
    // V^T "row 64": ones for the lanes that hold d = 0 of the extra block, zeros elsewhere
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)(c16 == 0 ? 1.0f : 0.0f);

    // LDS-DMA sources: one running pointer per piece, advanced by 64 keys per tile (no per-tile address arithmetic);
    // only a tile that reaches past S takes the checked form (rows beyond S read a zero line)
    const int lr = lane >> 3, pc = lane & 7;
    const bf16* dsrc[PIECES];
    long dstep[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int pce = wave * PIECES + i;
        const int isv = pce >> 3, row = (pce & 7) * 8 + lr;
        const int c = pc ^ (isv ? swz_v16(row) : swz_k16(row));
        dsrc[i] = isv ? Vb + (size_t)row * ldv + c * 8 : Kb + (size_t)row * ldk + c * 8;
        dstep[i] = (long)ATT_KV * (isv ? ldv : ldk);
    }
    auto dma_tile = [&](int kt, int buf) {           // tiles are issued in order: kt = 0, 1, 2, ...
        const bool tail = (kt + 1) * ATT_KV > S;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int pce = wave * PIECES + i;
            const int isv = pce >> 3, rb = pce & 7;
            const bf16* src = dsrc[i];
            if (tail && kt * ATT_KV + rb * 8 + lr >= S) src = zeros;
            __builtin_amdgcn_global_load_lds((att_gbl_cvoid_t*)src, (att_lds_void_t*)(lds + buf * BUF_B + isv * TILE_B + rb * 1024), 16, 0, 0);
            dsrc[i] += dstep[i];
        }
    };
    const int k_off0 = c16 * 128 + (((0 + g) ^ swz_k16(c16)) << 4);
    const int k_off1 = c16 * 128 + (((4 + g) ^ swz_k16(c16)) << 4);
    const int vkey = 4 * g + (c16 >> 2);
    const int vsw = swz_v16(vkey);
    const int vrow = vkey * 128 + 8 * (c16 & 1);
    const int vbit = (c16 & 3) >> 1;
    const int v_off0 = vrow + (((0 ^ vsw) + vbit) << 4), v_off1 = vrow + (((2 ^ vsw) + vbit) << 4);
    const int v_off2 = vrow + (((4 ^ vsw) + vbit) << 4), v_off3 = vrow + (((6 ^ vsw) + vbit) << 4);

    float m_ref = 0.f;                                // reference maximum of this lane's query row (base-2 exponent units)
    auto qk_tile = [&](int buf, f32x4 (&s)[4]) {
        const char* kb_ = lds + buf * BUF_B;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const bf16x8 ka = *reinterpret_cast<const bf16x8*>(kb_ + kb * 2048 + k_off0);
            const bf16x8 kc = *reinterpret_cast<const bf16x8*>(kb_ + kb * 2048 + k_off1);
            f32x4 acc = {-m_ref, -m_ref, -m_ref, -m_ref};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[0], acc, 0, 0, 0);
            s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc, qf[1], acc, 0, 0, 0);
        }
    };

    f32x4 o[5];                                       // O^T d blocks 0..3; o[4] row 0 = running row sum
#pragma unroll
    for (int i = 0; i < 5; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkt = (S + ATT_KV - 1) / ATT_KV;

    dma_tile(0, 0);
    if (nkt > 1) dma_tile(1, 1);
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 sa[4], sb[4];                               // scores of the current / next tile, trading places every trip
    qk_tile(0, sa);

    // one trip: tile kt (scores in `s`, V in buffer cur); leaves the scores of tile kt+1 in `sn`
    auto trip = [&](f32x4 (&s)[4], f32x4 (&sn)[4], int kt, int cur, int nb, int fb) {
#ifndef ST_ATT_NOSYNC     // (timing experiment: no DMA, no barrier - the loop re-reads whatever the ring holds)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own pieces of tile kt+1 have landed ...
        __builtin_amdgcn_s_barrier();                              // ... and everyone's; tile kt-1 is dead
        if (kt + 2 < nkt) dma_tile(kt + 2, fb);
#endif
        // V^T fragments of this tile [key pair-block kp][d block]: issued now, first used after the softmax
        bf16x8 vf[2][4];
        {
            const char* vb = lds + cur * BUF_B + TILE_B;
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                vf[kp][0] = v_frag(vb, v_off0 + kp * 4096, v_off0 + kp * 4096 + 2048);
                vf[kp][1] = v_frag(vb, v_off1 + kp * 4096, v_off1 + kp * 4096 + 2048);
                vf[kp][2] = v_frag(vb, v_off2 + kp * 4096, v_off2 + kp * 4096 + 2048);
                vf[kp][3] = v_frag(vb, v_off3 + kp * 4096, v_off3 + kp * 4096 + 2048);
            }
        }
        // scores of the next tile: the matrix pipe works on them under this softmax (after the last tile the ring
        // slot holds an old tile: computed all the same, never used)
        qk_tile(nb, sn);
        if ((kt + 1) * ATT_KV > S) {                               // mask the tail keys (only the last tile has any)
            const int kbase = kt * ATT_KV + 4 * g;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kbase + 16 * kb + r >= S) s[kb][r] = -INFINITY;
        }
        float mx = att_max3(s[0][0], s[0][1], s[0][2]);
        mx = att_max3(mx, s[0][3], s[1][0]);
        mx = att_max3(mx, s[1][1], s[1][2]);
        mx = att_max3(mx, s[1][3], s[2][0]);
        mx = att_max3(mx, s[2][1], s[2][2]);
        mx = att_max3(mx, s[2][3], s[3][0]);
        mx = att_max3(mx, s[3][1], s[3][2]);
        mx = fmaxf(mx, s[3][3]);
        if (kt == 0 || __any(mx > ATT_LAG)) {
            // exact path: the row maximum (over the four lanes that share the row) becomes the reference of every row
            // that is on its first tile or has outrun the lag; everything already expressed against the old reference
            // (O, the row sum, this tile's and the next tile's scores) moves by the same amount
            const float rmx = xmax32(xmax16(mx));
            const float delta = ((kt == 0 || rmx > ATT_LAG) && rmx > -INFINITY) ? rmx : 0.f;
            const float alpha = kt == 0 ? 1.f : fast_exp2(-delta);      // (nothing to rescale on the first tile; delta may be very negative there)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kb][r] -= delta; sn[kb][r] -= delta; }
#pragma unroll
            for (int db = 0; db < 5; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[db][r] *= alpha;
            m_ref += delta;
        }
#ifndef ST_ATT_NOEXP      // (timing experiment: skip the exponentials)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][r] = fast_exp2(s[kb][r]);
#endif
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            const bf16x8 pb = pack8(s[2 * kp][0], s[2 * kp][1], s[2 * kp][2], s[2 * kp][3],
                                    s[2 * kp + 1][0], s[2 * kp + 1][1], s[2 * kp + 1][2], s[2 * kp + 1][3]);
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kp][db], pb, o[db], 0, 0, 0);
            o[4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb, o[4], 0, 0, 0);
        }
    };
    int cur = 0;
    for (int kt = 0; kt < nkt; kt += 2) {
        const int b1 = cur == 2 ? 0 : cur + 1, b2 = b1 == 2 ? 0 : b1 + 1;
        trip(sa, sb, kt, cur, b1, b2);
        if (kt + 1 < nkt) trip(sb, sa, kt + 1, b1, b2, cur);
        cur = b2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // row sum: row 0 of the extra block lives in register 0 of the lanes with g == 0
    const float l = __shfl(o[4][0], c16, 64);
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

// ---- 32-row kernel, second generation (v_mfma_f32_32x32x16_bf16) -------------------------------------
// Same softmax economy as attn16v2_kernel (pre-scaled Q, accumulators started at -m_ref, lazy reference maximum,
// row sums from a third accumulator block fed with a V^T row of ones) on 32 query rows per wave: per row it issues
// half the MFMAs (a 32x32x16 holds the SIMD's issue port for 8 of its 32 cycles, a 16x16x32 for 8 of its 16), half
// the K / V fragment reads and half the LDS-DMA pieces - and at D = 64 the instruction issue of the SIMD, not the
// matrix pipe, is what the loop runs out of (measured: tools/build_one_variant.sh experiments).
// Pipeline, LDS images and the key split (KS) are those of attn_bf16_kernel.
template <int NW, int KS>
__global__ __launch_bounds__(NW * KS * 64) void attn32v2_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                                const bf16* __restrict__ V, bf16* __restrict__ O,
                                                                int T, int S, long ldq, long ldk, long ldv, long ldo, float scale_log2e) {
    constexpr int TILE_B = ATT_KV * 128;
    constexpr int BUF_B = 2 * TILE_B;
    constexpr int PIECES = 16 / NW > 0 ? 16 / NW : 1;
    static_assert(16 % NW == 0, "waves must divide the 16 DMA pieces of a tile");
    extern __shared__ __attribute__((aligned(16))) char lds_all[];      // KS rings of 3 tile buffers

    const int t_ = threadIdx.x, lane = t_ & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(t_ >> 6);
    const int kg = KS > 1 ? wave_all / NW : 0;
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

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 raw = *reinterpret_cast<const bf16x8*>(Qb + (size_t)qrow * ldq + 16 * ks + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (bf16)((float)raw[j] * scale_log2e);
    }
    bf16x8 ones;                                      // V^T "row 64" of the row-sum block: 1 for the lanes that hold its d = 0
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)(r32 == 0 ? 1.0f : 0.0f);

    const int nkt_all = (S + ATT_KV - 1) / ATT_KV;
    const int nkt = (nkt_all + KS - 1) / KS;
    const int kt0 = kg * nkt;

    // LDS-DMA sources: one running pointer per piece, advanced by 64 keys per tile; only a tile that reaches past S
    // takes the checked form (rows beyond S read a zero line)
    const int lr = lane >> 3, pc = lane & 7;
    const bf16* dsrc[PIECES];
    long dstep[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int pce = wave * PIECES + i;
        const int isv = pce >> 3, row = (pce & 7) * 8 + lr;
        const int c = pc ^ (isv ? swz_v(row) : swz_k(row));
        const size_t key0 = (size_t)kt0 * ATT_KV + row;
        dsrc[i] = isv ? Vb + key0 * ldv + c * 8 : Kb + key0 * ldk + c * 8;
        dstep[i] = (long)ATT_KV * (isv ? ldv : ldk);
    }
    auto dma_tile = [&](int kt, int buf) {           // tiles are issued in order: kt = 0, 1, 2, ...
        const bool tail = (kt0 + kt + 1) * ATT_KV > S;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int pce = wave * PIECES + i;
            const int isv = pce >> 3, rb = pce & 7;
            const bf16* src = dsrc[i];
            if (tail && (kt0 + kt) * ATT_KV + rb * 8 + lr >= S) src = zeros;
            __builtin_amdgcn_global_load_lds((att_gbl_cvoid_t*)src, (att_lds_void_t*)(lds + buf * BUF_B + isv * TILE_B + rb * 1024), 16, 0, 0);
            dsrc[i] += dstep[i];
        }
    };
    // lane-constant offsets of the transposed V reads (d block 0 / 1); k-step s_ adds 2048, the second key half 1024
    int v_base0, v_base1;
    {
        const int q4 = (lane & 15) >> 2;
        const int key = 4 * h + q4;
        const int ch0 = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
        const int row = key * 128 + 8 * (lane & 1);
        v_base0 = row + ((ch0 ^ swz_v(key)) << 4);
        v_base1 = row + (((ch0 + 4) ^ swz_v(key)) << 4);
    }
    float m_ref = 0.f;
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
        for (int r = 0; r < 16; ++r) { s0[r] = -m_ref; s1[r] = -m_ref; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[ks], qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[ks], qf[ks], s1, 0, 0, 0);
        }
    };

    f32x16 o0 = {0}, o1 = {0}, o2 = {0};              // O^T rows d 0..31, 32..63; o2 row 0 = running row sum
    dma_tile(0, 0);
    if (nkt > 1) dma_tile(1, 1);
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 sa0, sa1, sb0, sb1;                        // scores of the current / next tile, trading places every trip
    qk_tile(0, sa0, sa1);

    auto trip = [&](f32x16& s0, f32x16& s1, f32x16& n0, f32x16& n1, int kt, int cur, int nb, int fb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) dma_tile(kt + 2, fb);
        // V^T fragments of this tile [k-step][d block]: issued now, first used after the softmax
        bf16x8 vf[4][2];
        {
            const char* vb = lds + cur * BUF_B + TILE_B;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                vf[s_][0] = v_frag(vb, v_base0 + s_ * 2048, v_base0 + s_ * 2048 + 1024);
                vf[s_][1] = v_frag(vb, v_base1 + s_ * 2048, v_base1 + s_ * 2048 + 1024);
            }
        }
        qk_tile(nb, n0, n1);                                       // next tile's scores (after the last tile: computed on an old slot, unused)
        if ((kt0 + kt + 1) * ATT_KV > S) {
            const int kbase = (kt0 + kt) * ATT_KV + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= S) s0[r] = -INFINITY;
                if (key + 32 >= S) s1[r] = -INFINITY;
            }
        }
        float mx = att_max3(s0[0], s0[1], s0[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = att_max3(mx, s0[r], s0[r + 1]);
        mx = att_max3(mx, s0[15], s1[0]);
#pragma unroll
        for (int r = 1; r < 15; r += 2) mx = att_max3(mx, s1[r], s1[r + 1]);
        mx = fmaxf(mx, s1[15]);
        if (kt == 0 || __any(mx > ATT_LAG)) {
            // exact path (first tile, or a row outran the lag): see attn16v2_kernel
            const float rmx = xmax32(mx);
            const float delta = ((kt == 0 || rmx > ATT_LAG) && rmx > -INFINITY) ? rmx : 0.f;
            const float alpha = kt == 0 ? 1.f : fast_exp2(-delta);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] -= delta; s1[r] -= delta; n0[r] -= delta; n1[r] -= delta;
                o0[r] *= alpha; o1[r] *= alpha; o2[r] *= alpha;
            }
            m_ref += delta;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = fast_exp2(s0[r]); s1[r] = fast_exp2(s1[r]); }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const f32x16& sx = s_ < 2 ? s0 : s1;
            const int e = 8 * (s_ & 1);
            const bf16x8 pb = pack8(sx[e], sx[e + 1], sx[e + 2], sx[e + 3], sx[e + 4], sx[e + 5], sx[e + 6], sx[e + 7]);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s_][0], pb, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s_][1], pb, o1, 0, 0, 0);
            o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pb, o2, 0, 0, 0);
        }
    };
    int cur = 0;
    for (int kt = 0; kt < nkt; kt += 2) {
        const int b1 = cur == 2 ? 0 : cur + 1, b2 = b1 == 2 ? 0 : b1 + 1;
        trip(sa0, sa1, sb0, sb1, kt, cur, b1, b2);
        if (kt + 1 < nkt) trip(sb0, sb1, sa0, sa1, kt + 1, b1, b2, cur);
        cur = b2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // row sum of query r32: register 0 of the lanes with h == 0 (row 0 of the third block)
    float l = __shfl(o2[0], r32, 64);
    if constexpr (KS > 1) {
        // merge the key groups: group 1 parks (O, m_ref, l) in LDS (the rings are dead), group 0 folds them in
        __syncthreads();
        float* park = reinterpret_cast<float*>(lds_all) + (size_t)wave * (34 * 64) + lane;
        if (kg == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { park[r * 64] = o0[r]; park[(16 + r) * 64] = o1[r]; }
            park[32 * 64] = m_ref; park[33 * 64] = l;
        }
        __syncthreads();
        if (kg == 1) return;
        const float m2 = park[32 * 64], l2 = park[33 * 64];
        const float m_new = fmaxf(m_ref, m2);
        const float a1 = fast_exp2(m_ref - m_new), a2 = fast_exp2(m2 - m_new);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] = o0[r] * a1 + park[r * 64] * a2;
            o1[r] = o1[r] * a1 + park[(16 + r) * 64] * a2;
        }
        l = l * a1 + l2 * a2;
    }
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

// ---- 32-row kernel, staggered: eight waves whose two halves take turns on the matrix pipe -----------
// In attn32v2_kernel the two waves of a SIMD belong to one block and meet at the same barrier every tile, so they run
// in step: both multiply at once (fighting over the pipe), both do softmax at once (pipe idle).  Here a trip is two
// phases separated by block barriers,
//     V: the softmax of this tile's scores -> bf16 P fragments                       (VALU only)
//     M: twelve P V MFMAs, then the eight K Q^T MFMAs of the next tile, and between them the LDS reads of the NEXT
//        trip's fragments (V^T of tile k+1 into the registers the P V MFMAs have just consumed, K of tile k+2 likewise)
//        and the DMA issue of tile k+4 - the loads land while the wave is in its next V phase
// and waves 4-7 - the second wave of every SIMD - run one phase behind waves 0-3 (one extra barrier up front, repaid
// at the end): while one half multiplies, the other half's softmax runs on the VALU of the same SIMD.  Measured with
// in-kernel stamps (tools/att_probe2.py): M = 690 cycles; a V phase that also carried the fragment reads and the DMA
// issue took 2150 (one wave cannot hide its own LDS / DMA latencies), the softmax alone 790.
// Eight row waves (256 query rows) share one ring of six K/V tiles.  Tile t is issued at the top of V(t-4), retired by
// the issuing wave's counted vmcnt at the end of V(t-3) (one DMA group of its own is younger: a full trip of flight
// time), first read - K fragments - in M(t-2): for the early half that is two barriers after the late half retired its
// pieces.  Fragment reads are retired (lgkmcnt(0)) before the barrier that ends a V phase, so a slot is free for the
// DMA of tile t+6 two phases after the late half's last read of tile t.
template <bool STG>
__global__ __launch_bounds__(512) void attn32s_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                      const bf16* __restrict__ V, bf16* __restrict__ O,
                                                      int T, int S, long ldq, long ldk, long ldv, long ldo, float scale_log2e,
                                                      unsigned long long* probe) {
    constexpr int NW = 8;
    constexpr int TILE_B = ATT_KV * 128;
    constexpr int BUF_B = 2 * TILE_B;
    constexpr int RB = 6;                             // ring buffers
    constexpr int PIECES = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) char lds[];      // the ring, then a 1-KiB dump for the dummy DMAs
    char* const dump = lds + RB * BUF_B;
#ifdef ST_PROBE
    unsigned long long pv = 0, pvw = 0, pm = 0, pmw = 0;
#endif

    const int t_ = threadIdx.x, lane = t_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t_ >> 6);
    const int late = wave >> 2;                       // 1: this wave runs one phase behind
    const int r32 = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int qrow = min(q0 + r32, T - 1);

    const bf16* Qb = Q + (size_t)b * T * ldq + (size_t)head * ATT_D;
    const bf16* Kb = K + (size_t)b * S * ldk + (size_t)head * ATT_D;
    const bf16* Vb = V + (size_t)b * S * ldv + (size_t)head * ATT_D;
    const bf16* zeros = reinterpret_cast<const bf16*>(g_att_zero16);

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 raw = *reinterpret_cast<const bf16x8*>(Qb + (size_t)qrow * ldq + 16 * ks + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (bf16)((float)raw[j] * scale_log2e);
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)(r32 == 0 ? 1.0f : 0.0f);

    const int nkt = (S + ATT_KV - 1) / ATT_KV;
    const int lr = lane >> 3, pc = lane & 7;
    const bf16* dsrc[PIECES];
    long dstep[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int pce = wave * PIECES + i;
        const int isv = pce >> 3, row = (pce & 7) * 8 + lr;
        const int c = pc ^ (isv ? swz_v(row) : swz_k(row));
        dsrc[i] = isv ? Vb + (size_t)row * ldv + c * 8 : Kb + (size_t)row * ldk + c * 8;
        dstep[i] = (long)ATT_KV * (isv ? ldv : ldk);
    }
    // tiles are issued in order kt = 0, 1, 2, ...; past the last tile the pieces become dummies (one zero line into the
    // dump area) so that every trip issues the same number of DMAs and the counted waits stay valid
    int dbuf = 0;                                     // ring slot of the next tile to issue
    auto dma_tile = [&](int kt) {
        char* const slot = lds + dbuf * BUF_B;
        if ((kt + 1) * ATT_KV <= S) {                 // a whole tile inside S: no per-lane checks, three instructions per piece
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int pce = wave * PIECES + i;
                __builtin_amdgcn_global_load_lds((att_gbl_cvoid_t*)dsrc[i], (att_lds_void_t*)(slot + (pce >> 3) * TILE_B + (pce & 7) * 1024), 16, 0, 0);
            }
        } else {                                      // the tile that reaches past S, and the dummies after the last tile
            const bool live = kt < nkt;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int pce = wave * PIECES + i;
                const int isv = pce >> 3, rb = pce & 7;
                const bf16* src = (live && kt * ATT_KV + rb * 8 + lr < S) ? dsrc[i] : zeros;
                __builtin_amdgcn_global_load_lds((att_gbl_cvoid_t*)src, (att_lds_void_t*)(live ? slot + isv * TILE_B + rb * 1024 : dump), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < PIECES; ++i) dsrc[i] += dstep[i];
        dbuf = dbuf == RB - 1 ? 0 : dbuf + 1;
    };
    // STG: the pieces travel through registers instead (global_load a trip ahead, ds_write at the top of the next V
    // phase) - a DMA instruction costs the issuing wave ~200 cycles, a global_load + ds_write_b128 pair a few tens
    typedef unsigned int stg_t __attribute__((ext_vector_type(4)));
    stg_t stg[PIECES];
    auto load_tile = [&](int kt) {                    // tile kt -> registers
        if (kt < nkt) {
            const bool whole = (kt + 1) * ATT_KV <= S;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int rb = (wave * PIECES + i) & 7;
                const bf16* src = (whole || kt * ATT_KV + rb * 8 + lr < S) ? dsrc[i] : zeros;
                stg[i] = *reinterpret_cast<const stg_t*>(src);
                dsrc[i] += dstep[i];
            }
        }
    };
    auto store_tile = [&](int kt) {                   // registers -> ring slot of tile kt
        if (kt < nkt) {
            char* const slot = lds + dbuf * BUF_B;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int pce = wave * PIECES + i;
                *reinterpret_cast<stg_t*>(slot + (pce >> 3) * TILE_B + (pce & 7) * 1024 + lane * 16) = stg[i];
            }
        }
        dbuf = dbuf == RB - 1 ? 0 : dbuf + 1;
    };
    int v_base0, v_base1;
    {
        const int q4 = (lane & 15) >> 2;
        const int key = 4 * h + q4;
        const int ch0 = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
        const int row = key * 128 + 8 * (lane & 1);
        v_base0 = row + ((ch0 ^ swz_v(key)) << 4);
        v_base1 = row + (((ch0 + 4) ^ swz_v(key)) << 4);
    }
    int k_off[4][2];                                  // K fragment offsets [k-step][key half]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int c = 2 * ks + h;
        k_off[ks][0] = r32 * 128 + ((c ^ swz_k(r32)) << 4);
        k_off[ks][1] = (32 + r32) * 128 + ((c ^ swz_k(32 + r32)) << 4);
    }
    float m_ref = 0.f;
    f32x16 o0 = {0}, o1 = {0}, o2 = {0};
    bf16x8 kf0[4], kf1[4], vf[4][2], pb[4];

    // prologue: four tiles in flight, the first three landed; scores of tile 0; fragments of the first M phase
    dma_tile(0); dma_tile(1); dma_tile(2);
    if constexpr (STG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        load_tile(3);
    } else {
        dma_tile(3);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    f32x16 sa0, sa1, sb0, sb1;
    {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf0[ks] = *reinterpret_cast<const bf16x8*>(lds + k_off[ks][0]);
            kf1[ks] = *reinterpret_cast<const bf16x8*>(lds + k_off[ks][1]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa0[r] = 0.f; sa1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            sa0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[ks], qf[ks], sa0, 0, 0, 0);
            sa1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[ks], qf[ks], sa1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {                 // V^T of tile 0, K of tile 1
            vf[s_][0] = v_frag(lds + TILE_B, v_base0 + s_ * 2048, v_base0 + s_ * 2048 + 1024);
            vf[s_][1] = v_frag(lds + TILE_B, v_base1 + s_ * 2048, v_base1 + s_ * 2048 + 1024);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf0[ks] = *reinterpret_cast<const bf16x8*>(lds + BUF_B + k_off[ks][0]);
            kf1[ks] = *reinterpret_cast<const bf16x8*>(lds + BUF_B + k_off[ks][1]);
        }
    }
    if (late) __builtin_amdgcn_s_barrier();          // the second half runs one phase behind the first

    int vslot = 1, kslot = 2;                         // ring slots of tile kt+1 (V^T) and tile kt+2 (K)
    auto trip = [&](f32x16& s0, f32x16& s1, f32x16& n0, f32x16& n1, int kt) {
        // ---- phase V: DMA issue of tile kt+4, softmax of tile kt ----
        AP_STAMP(t0)
        if constexpr (STG) {
            store_tile(kt + 3);
            load_tile(kt + 4);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            dma_tile(kt + 4);
        }
        if ((kt + 1) * ATT_KV > S) {
            const int kbase = kt * ATT_KV + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= S) s0[r] = -INFINITY;
                if (key + 32 >= S) s1[r] = -INFINITY;
            }
        }
        // four independent maximum chains (one dependent chain of sixteen v_max3 is all latency)
        float ma = att_max3(s0[0], s0[1], s0[2]), mb = att_max3(s0[8], s0[9], s0[10]);
        float mc = att_max3(s1[0], s1[1], s1[2]), md = att_max3(s1[8], s1[9], s1[10]);
        ma = att_max3(ma, s0[3], s0[4]); mb = att_max3(mb, s0[11], s0[12]); mc = att_max3(mc, s1[3], s1[4]); md = att_max3(md, s1[11], s1[12]);
        ma = att_max3(ma, s0[5], s0[6]); mb = att_max3(mb, s0[13], s0[14]); mc = att_max3(mc, s1[5], s1[6]); md = att_max3(md, s1[13], s1[14]);
        ma = fmaxf(ma, s0[7]); mb = fmaxf(mb, s0[15]); mc = fmaxf(mc, s1[7]); md = fmaxf(md, s1[15]);
        const float mx = fmaxf(fmaxf(ma, mb), fmaxf(mc, md));
        if (kt == 0 || __any(mx > ATT_LAG)) {
            // exact path (first tile, or a row outran the lag): see attn16v2_kernel; the next tile's scores do not exist yet
            const float rmx = xmax32(mx);
            const float delta = ((kt == 0 || rmx > ATT_LAG) && rmx > -INFINITY) ? rmx : 0.f;
            const float alpha = kt == 0 ? 1.f : fast_exp2(-delta);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] -= delta; s1[r] -= delta;
                o0[r] *= alpha; o1[r] *= alpha; o2[r] *= alpha;
            }
            m_ref += delta;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = fast_exp2(s0[r]); s1[r] = fast_exp2(s1[r]); }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const f32x16& sx = s_ < 2 ? s0 : s1;
            const int e = 8 * (s_ & 1);
            pb[s_] = pack8(sx[e], sx[e + 1], sx[e + 2], sx[e + 3], sx[e + 4], sx[e + 5], sx[e + 6], sx[e + 7]);
        }
        __builtin_amdgcn_sched_barrier(0);
        AP_STAMP(t1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the fragment reads issued in the last M phase are in registers
        if constexpr (!STG)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");   // own pieces of tile kt+3 (issued a trip ago) have landed; tile kt+4's may fly
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase M: the matrix pipe, with the next trip's loads between the MFMAs ----
        AP_STAMP(t2)
        {
            const char* vb = lds + vslot * BUF_B + TILE_B;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s_][0], pb[s_], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s_][1], pb[s_], o1, 0, 0, 0);
                o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pb[s_], o2, 0, 0, 0);
                vf[s_][0] = v_frag(vb, v_base0 + s_ * 2048, v_base0 + s_ * 2048 + 1024);      // V^T of tile kt+1
                vf[s_][1] = v_frag(vb, v_base1 + s_ * 2048, v_base1 + s_ * 2048 + 1024);
            }
            const char* kb = lds + kslot * BUF_B;
#pragma unroll
            for (int r = 0; r < 16; ++r) { n0[r] = -m_ref; n1[r] = -m_ref; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[ks], qf[ks], n0, 0, 0, 0);
                n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[ks], qf[ks], n1, 0, 0, 0);
                kf0[ks] = *reinterpret_cast<const bf16x8*>(kb + k_off[ks][0]);                 // K of tile kt+2
                kf1[ks] = *reinterpret_cast<const bf16x8*>(kb + k_off[ks][1]);
            }
        }
        vslot = vslot == RB - 1 ? 0 : vslot + 1;
        kslot = kslot == RB - 1 ? 0 : kslot + 1;
        __builtin_amdgcn_sched_barrier(0);
        AP_STAMP(t3)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        AP_STAMP(t4)
        AP_ADD(pv, t1, t0) AP_ADD(pvw, t2, t1) AP_ADD(pm, t3, t2) AP_ADD(pmw, t4, t3)
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        trip(sa0, sa1, sb0, sb1, kt);
        if (kt + 1 < nkt) trip(sb0, sb1, sa0, sa1, kt + 1);
    }
#ifdef ST_PROBE
    if (probe && lane == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        unsigned long long* o = probe + wave * 8;
        o[0] = pv; o[1] = pvw; o[2] = pm; o[3] = pmw; o[4] = nkt; o[5] = 0; o[6] = 0; o[7] = 0;
    }
#endif
    if (!late) __builtin_amdgcn_s_barrier();         // barrier counts of the two halves are equal again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    const float l = __shfl(o2[0], r32, 64);
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

// ---- 32-row kernel, interleaved: softmax instructions placed in the gaps between the MFMAs of the same wave ----------
// At D = 64 a 64-key tile costs a wave 20 MFMAs (640 pipe cycles) and ~80 VALU instructions, 32 of them v_exp_f32 at
// 8 issue cycles: ~420 cycles of vector issue, which fit the 24 free issue cycles of each MFMA gap only if they are
// PLACED there (a wave issues in order: eight MFMAs in a row stall it at the second one, and the VALU behind them
// waits).  One trip of this kernel is a single scheduling region, pinned gap by gap with sched_barrier:
//     QK phase, 8 MFMAs  K(t+1) Q^T -> next scores   | exp + pack of the first half of tile t | V^T(t) fragment reads
//     PV phase, 12 MFMAs V^T(t) P(t) (+ ones block)  | exp + pack of the second half, max of the next scores | K(t+2) reads
// then the counted DMA wait, one block barrier, and the (rare) branch that moves the lazy maximum.  Because P(t) is
// multiplied in the trip that exponentiates it, nothing is pending when the maximum moves.
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn32i_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                      const bf16* __restrict__ V, bf16* __restrict__ O,
                                                      int T, int S, long ldq, long ldk, long ldv, long ldo, float scale_log2e,
                                                      unsigned long long* probe) {
    constexpr int TILE_B = ATT_KV * 128;
    constexpr int BUF_B = 2 * TILE_B;
    constexpr int RB = 6;                             // ring buffers
    constexpr int PIECES = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) char lds[];      // the ring, then a 1-KiB dump for the dummy DMAs
#ifdef ST_PROBE
    unsigned long long pv = 0, pvw = 0, pm = 0, pmw = 0;
#endif

    const int t_ = threadIdx.x, lane = t_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t_ >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int qrow = min(q0 + r32, T - 1);

    const bf16* Qb = Q + (size_t)b * T * ldq + (size_t)head * ATT_D;
    const bf16* Kb = K + (size_t)b * S * ldk + (size_t)head * ATT_D;
    const bf16* Vb = V + (size_t)b * S * ldv + (size_t)head * ATT_D;
    const bf16* zeros = reinterpret_cast<const bf16*>(g_att_zero16);

    // Q through asm loads: a compiler-counted wait would not know about the DMAs issued behind them and would drain those too
    typedef unsigned int q_raw_t __attribute__((ext_vector_type(4)));
    q_raw_t qraw[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(qraw[ks]) : "v"(Qb + (size_t)qrow * ldq + 16 * ks + 8 * h) : "memory");
    bf16x8 qf[4];
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)(r32 == 0 ? 1.0f : 0.0f);

    const int nkt = (S + ATT_KV - 1) / ATT_KV;
    const int lr = lane >> 3, pc = lane & 7;
    const bf16* dsrc[PIECES];
    long dstep[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int pce = wave * PIECES + i;
        const int isv = pce >> 3, row = (pce & 7) * 8 + lr;
        const int c = pc ^ (isv ? swz_v(row) : swz_k(row));
        dsrc[i] = isv ? Vb + (size_t)row * ldv + c * 8 : Kb + (size_t)row * ldk + c * 8;
        dstep[i] = (long)ATT_KV * (isv ? ldv : ldk);
    }
    // tiles are issued in order; past the last tile the pieces become dummies (a zero line into the dump area) so that
    // every trip issues the same number of DMAs and the counted waits stay valid
    int dbuf = 0;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(att_lds_offset(lds));
    const unsigned dump_off = lds0 + RB * BUF_B;
    auto dma_piece = [&](int kt, int i) {             // piece i of tile kt into ring slot dbuf
        const unsigned slot = lds0 + dbuf * BUF_B;
        const int pce = wave * PIECES + i;
        const int isv = pce >> 3, rb = pce & 7;
        if ((kt + 1) * ATT_KV <= S) {
            att_dma16(dsrc[i], slot + isv * TILE_B + rb * 1024);
        } else {
            const bool live = kt < nkt;
            const bf16* src = (live && kt * ATT_KV + rb * 8 + lr < S) ? dsrc[i] : zeros;
            att_dma16(src, live ? slot + isv * TILE_B + rb * 1024 : dump_off);
        }
        dsrc[i] += dstep[i];
    };
    auto dma_next = [&]() { dbuf = dbuf == RB - 1 ? 0 : dbuf + 1; };
    auto dma_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) dma_piece(kt, i);
        dma_next();
    };
    int v_base0, v_base1;
    {
        const int q4 = (lane & 15) >> 2;
        const int key = 4 * h + q4;
        const int ch0 = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
        const int row = key * 128 + 8 * (lane & 1);
        v_base0 = row + ((ch0 ^ swz_v(key)) << 4);
        v_base1 = row + (((ch0 + 4) ^ swz_v(key)) << 4);
    }
    int k_off[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int c = 2 * ks + h;
        k_off[ks][0] = r32 * 128 + ((c ^ swz_k(r32)) << 4);
        k_off[ks][1] = (32 + r32) * 128 + ((c ^ swz_k(32 + r32)) << 4);
    }
    f32x16 negm;                                       // -m_ref of this lane's query row in every register: the C operand of a tile's first MFMAs
    f32x16 o0 = {0}, o1 = {0}, o2 = {0};
    bf16x8 kf0[4], kf1[4], vf[4][2], pb[4];
    auto mask_tail = [&](f32x16& s0, f32x16& s1, int kt) {
        const int kbase = kt * ATT_KV + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kbase + (r & 3) + 8 * (r >> 2);
            if (key >= S) s0[r] = -INFINITY;
            if (key + 32 >= S) s1[r] = -INFINITY;
        }
    };
    auto max32 = [&](const f32x16& s0, const f32x16& s1) {
        float ma = att_max3(s0[0], s0[1], s0[2]), mb = att_max3(s0[8], s0[9], s0[10]);
        float mc = att_max3(s1[0], s1[1], s1[2]), md = att_max3(s1[8], s1[9], s1[10]);
        ma = att_max3(ma, s0[3], s0[4]); mb = att_max3(mb, s0[11], s0[12]); mc = att_max3(mc, s1[3], s1[4]); md = att_max3(md, s1[11], s1[12]);
        ma = att_max3(ma, s0[5], s0[6]); mb = att_max3(mb, s0[13], s0[14]); mc = att_max3(mc, s1[5], s1[6]); md = att_max3(md, s1[13], s1[14]);
        ma = att_max3(ma, s0[7], mb); mc = att_max3(mc, s1[7], md);
        return att_max3(ma, s0[15], att_max3(mc, s1[15], mc));
    };

    // prologue: four tiles in flight, the first three landed; exact scores of tile 0; K fragments of tile 1
    dma_tile(0); dma_tile(1); dma_tile(2); dma_tile(3);
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(qraw[0]), "+v"(qraw[1]), "+v"(qraw[2]), "+v"(qraw[3]) : "n"(3 * PIECES) : "memory");   // Q and tile 0: start on them while the others fly
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 raw = __builtin_bit_cast(bf16x8, qraw[ks]);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (bf16)((float)raw[j] * scale_log2e);
    }
    f32x16 sa0, sa1, sb0, sb1;
    {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf0[ks] = *reinterpret_cast<const bf16x8*>(lds + k_off[ks][0]);
            kf1[ks] = *reinterpret_cast<const bf16x8*>(lds + k_off[ks][1]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa0[r] = 0.f; sa1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            sa0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[ks], qf[ks], sa0, 0, 0, 0);
            sa1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[ks], qf[ks], sa1, 0, 0, 0);
        }
        if (ATT_KV > S) mask_tail(sa0, sa1, 0);
        const float rmx = xmax32(max32(sa0, sa1));
        const float m0 = rmx > -INFINITY ? rmx : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa0[r] -= m0; sa1[r] -= m0; negm[r] = -m0; }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf0[ks] = *reinterpret_cast<const bf16x8*>(lds + BUF_B + k_off[ks][0]);
            kf1[ks] = *reinterpret_cast<const bf16x8*>(lds + BUF_B + k_off[ks][1]);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");           // tile 2: its K fragments are read in the first trip
        __builtin_amdgcn_s_barrier();
    }

    int vslot = 0, kslot = 2;                         // ring slots of tile kt (V^T) and tile kt+2 (K)
#ifdef ATT_I_PRIO
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    auto trip = [&](f32x16& s0, f32x16& s1, f32x16& n0, f32x16& n1, int kt) {
        AP_STAMP(t0)
        AP_STAMP(t1)
        const char* vb = lds + vslot * BUF_B + TILE_B;
        const char* kb = lds + kslot * BUF_B;
        float mch[4];                                  // four maximum chains over the next tile's scores
        // ---- QK phase: gap g carries MFMA g, two exponentials + their pack, one V^T fragment (two transposed reads) ----
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int ks = g >> 1;
            if ((g & 1) == 0) n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[ks], qf[ks], ks == 0 ? negm : n0, 0, 0, 0);
            else              n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[ks], qf[ks], ks == 0 ? negm : n1, 0, 0, 0);
#ifdef ATT_I_NOEXP
            const float e0 = s0[2 * g], e1 = s0[2 * g + 1];
#else
            const float e0 = fast_exp2(s0[2 * g]), e1 = fast_exp2(s0[2 * g + 1]);
#endif
            pb[g >> 2][2 * (g & 3)] = (bf16)e0; pb[g >> 2][2 * (g & 3) + 1] = (bf16)e1;
#if defined(ATT_I_DMA_SPREAD) && !defined(ATT_I_NODMA)
            if (PIECES == 2 ? g == 3 : (g == 1 || g == 5)) dma_piece(kt + 4, PIECES == 2 ? 0 : (g == 1 ? 0 : 1));
#endif
            vf[g >> 1][g & 1] = v_frag(vb, ((g & 1) ? v_base1 : v_base0) + (g >> 1) * 2048, ((g & 1) ? v_base1 : v_base0) + (g >> 1) * 2048 + 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        AP_STAMP(tq)
        // ---- PV phase: gap p carries MFMA p; gaps 0-7 the other sixteen exponentials, gaps 2-11 the maximum of the next
        //      tile's scores, gaps 4-11 the K fragments of tile kt+2 ----
#pragma unroll
        for (int p = 0; p < 12; ++p) {
            const int s_ = p / 3, w = p % 3;
            if (w == 0)      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s_][0], pb[s_], o0, 0, 0, 0);
            else if (w == 1) o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s_][1], pb[s_], o1, 0, 0, 0);
            else             o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pb[s_], o2, 0, 0, 0);
            if (p < 8) {
#ifdef ATT_I_NOEXP
                const float e0 = s1[2 * p], e1 = s1[2 * p + 1];
#else
                const float e0 = fast_exp2(s1[2 * p]), e1 = fast_exp2(s1[2 * p + 1]);
#endif
                pb[2 + (p >> 2)][2 * (p & 3)] = (bf16)e0; pb[2 + (p >> 2)][2 * (p & 3) + 1] = (bf16)e1;
            }
#ifdef ATT_I_NOMAX
            if (p == 2) { mch[0] = n0[0]; mch[1] = n0[8]; mch[2] = n1[0]; mch[3] = n1[8]; }
#define ATT_I_MAXOFF && false
#else
#define ATT_I_MAXOFF
#endif
            if (p >= 2 && p < 6 ATT_I_MAXOFF) {                     // one chain start per gap
                const int c = p - 2;
                const f32x16& sx = c < 2 ? n0 : n1;
                const int e = 8 * (c & 1);
                mch[c] = att_max3(sx[e], sx[e + 1], sx[e + 2]);
            }
            if (p >= 6 && p < 8 ATT_I_MAXOFF) {
#pragma unroll
                for (int c = 2 * (p - 6); c < 2 * (p - 6) + 2; ++c) {
                    const f32x16& sx = c < 2 ? n0 : n1;
                    const int e = 8 * (c & 1);
                    mch[c] = att_max3(mch[c], sx[e + 3], sx[e + 4]);
                }
            }
            if (p >= 8 ATT_I_MAXOFF) {
                const int c = p - 8;
                const f32x16& sx = c < 2 ? n0 : n1;
                const int e = 8 * (c & 1);
                mch[c] = att_max3(mch[c], sx[e + 5], sx[e + 6]);
                mch[c] = att_max3(mch[c], sx[e + 7], sx[e + 7]);
            }
#ifndef ATT_I_NODMA
            // LDS-DMA of tile kt+4 (spreading the pieces over the gaps by wave was tried: the per-gap branches cost more
            // than the queueing of sixteen simultaneous wave-instructions in the CU's load path)
#if defined(ATT_I_DMA_SPREAD)
            if (PIECES == 2 ? p == 9 : (p == 3 || p == 9)) dma_piece(kt + 4, PIECES == 2 ? 1 : (p == 3 ? 2 : 3));
#elif defined(ATT_I_DMA_SPREAD2)
            if (PIECES == 2 ? (p == 2 || p == 8) : (p == 1 || p == 4 || p == 7 || p == 10)) dma_piece(kt + 4, PIECES == 2 ? (p == 8) : (p - 1) / 3);
#elif !defined(ATT_I_DMA_END)
            if (p >= 12 - PIECES) dma_piece(kt + 4, p - (12 - PIECES));
#endif
#endif
#ifdef ATT_I_NOLDS
            if (kt == 0)
#endif
            if (p >= 4) {
                const int i = p - 4, ks = i >> 1;
                if ((i & 1) == 0) kf0[ks] = *reinterpret_cast<const bf16x8*>(kb + k_off[ks][0]);
                else              kf1[ks] = *reinterpret_cast<const bf16x8*>(kb + k_off[ks][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float mx = att_max3(mch[0], mch[1], att_max3(mch[2], mch[3], mch[3]));
        vslot = vslot == RB - 1 ? 0 : vslot + 1;
        kslot = kslot == RB - 1 ? 0 : kslot + 1;
        AP_STAMP(t2)
        if (kt + 1 < nkt) {
            if ((kt + 2) * ATT_KV > S) { mask_tail(n0, n1, kt + 1); mx = max32(n0, n1); }
            if (__any(mx > ATT_LAG)) {
                // a row outran the lag: move its reference maximum (nothing is pending: P of this trip is already in O)
                const float rmx = xmax32(mx);
                const float delta = (rmx > ATT_LAG && rmx > -INFINITY) ? rmx : 0.f;
                const float alpha = fast_exp2(-delta);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    n0[r] -= delta; n1[r] -= delta; negm[r] -= delta;
                    o0[r] *= alpha; o1[r] *= alpha; o2[r] *= alpha;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        AP_STAMP(t3)
#ifdef ATT_I_NODMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
#ifdef ATT_I_DMA_END
#pragma unroll
        for (int i = 0; i < PIECES; ++i) dma_piece(kt + 4, i);
#endif
#ifdef ATT_X_VM0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");   // own pieces of tile kt+3 have landed; tile kt+4's may fly
#endif
#ifdef ATT_X_LGKM0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#endif
        dma_next();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        AP_STAMP(t4)
        AP_ADD(pv, tq, t1) AP_ADD(pvw, t2, tq) AP_ADD(pm, t3, t2) AP_ADD(pmw, t4, t3)
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        trip(sa0, sa1, sb0, sb1, kt);
        if (kt + 1 < nkt) trip(sb0, sb1, sa0, sa1, kt + 1);
    }
#ifdef ST_PROBE
    if (probe && lane == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        unsigned long long* o = probe + wave * 8;
        o[0] = pv; o[1] = pvw; o[2] = pm; o[3] = pmw; o[4] = nkt; o[5] = 0; o[6] = 0; o[7] = 0;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    const float l = __shfl(o2[0], r32, 64);
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
        const int rows16 = rows16_env >= 0 ? rows16_env : (S < 256 ? 4 : 0);
        static const int v2 = att_dev_env_int("ST_ATT_V2", 1);
        if (rows16 && v2) {
            // second-generation 16-row kernel; the text-context launches get their own instantiation (kernel name)
            const bool cross = S <= 256;
#define ST_ATT16V2(NW_, TAG_) hipLaunchKernelGGL((attn16v2_kernel<NW_, TAG_>), dim3(cdiv(T, 16 * NW_), H, B), dim3(64 * NW_), 0, st, (const bf16*)q, \
                                                 (const bf16*)k, (const bf16*)v, (bf16*)out, T, S, ldq, ldk, ldv, ldo, c)
            if (rows16 == 8) { if (cross) ST_ATT16V2(8, 1); else ST_ATT16V2(8, 0); }
            else { if (cross) ST_ATT16V2(4, 1); else ST_ATT16V2(4, 0); }
#undef ST_ATT16V2
            return st_check_launch("attention");
        }
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
        // dev knob: 0 = interleaved kernel (product), 1/2 = staggered experiments, 5 = previous generation (attn32v2)
        static const int stag = att_dev_env_int("ST_ATT_STAG", 0);
        if (stag != 5 && S >= 256) {
            constexpr size_t RING5 = 6 * 2 * ATT_KV * 128 + 1024;
            if (stag == 0 || stag == 3 || stag == 4) {
                // eight waves (256 query rows) share a K/V ring unless that leaves half the CUs idle: SDXL's 32x32 level at
                // batch 1 is 80 such blocks; as 160 blocks of four waves every SIMD holds one wave (16.9 us against 20.4)
                const int nwi = stag == 3 ? 8 : stag == 4 ? 4 : ((long)cdiv(T, 256) * H * B <= 128 ? 4 : 8);
                auto kfi = nwi == 8 ? attn32i_kernel<8> : attn32i_kernel<4>;
                static bool o3_ = (hipFuncSetAttribute((const void*)attn32i_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RING5),
                                   hipFuncSetAttribute((const void*)attn32i_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RING5), true);
                (void)o3_;
                hipLaunchKernelGGL(kfi, dim3(cdiv(T, 32 * nwi), H, B), dim3(64 * nwi), RING5, st, (const bf16*)q, (const bf16*)k, (const bf16*)v,
                                   (bf16*)out, T, S, ldq, ldk, ldv, ldo, c, ATT_PROBE_ARG);
                return st_check_launch("attention");
            }
            auto kfs = stag == 2 ? attn32s_kernel<true> : attn32s_kernel<false>;
            static bool o2_ = (hipFuncSetAttribute((const void*)attn32s_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RING5),
                               hipFuncSetAttribute((const void*)attn32s_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RING5), true);
            (void)o2_;
            hipLaunchKernelGGL(kfs, dim3(cdiv(T, 256), H, B), dim3(512), RING5, st, (const bf16*)q, (const bf16*)k, (const bf16*)v,
                               (bf16*)out, T, S, ldq, ldk, ldv, ldo, c, ATT_PROBE_ARG);
            return st_check_launch("attention");
        }
        static const int v32 = att_dev_env_int("ST_ATT_32V2", 1);
        if (v32) {
#define ST_ATT32V2(NW_, KS_, GX_)                                                                                                   \
    do {                                                                                                                           \
        auto kfn2 = attn32v2_kernel<NW_, KS_>;                                                                                      \
        static bool once2 = (hipFuncSetAttribute((const void*)kfn2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KS_ * RING)), true); \
        (void)once2;                                                                                                               \
        hipLaunchKernelGGL(kfn2, dim3(GX_, H, B), dim3(64 * NW_ * KS_), KS_ * RING, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, \
                           (bf16*)out, T, S, ldq, ldk, ldv, ldo, c);                                                                \
    } while (0)
            if (split) ST_ATT32V2(4, 2, cdiv(T, 128));
            else if (nw == 8) ST_ATT32V2(8, 1, cdiv(T, 256));
            else ST_ATT32V2(4, 1, cdiv(T, 128));
#undef ST_ATT32V2
            return st_check_launch("attention");
        }
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
