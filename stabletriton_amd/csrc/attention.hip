// Fused attention core for gfx950: out = softmax(q k^T * scale) v, per head, no
// mask, head_dim 64 (every SDXL-base/refiner head).  Row A of SURVEY.md 8a:
// replaces the eager matmul/softmax/matmul of unet_pt.py:133-142 without ever
// materialising the (B,H,T,S) score tensor.
//
// bf16 kernels (flash style):
//   * attn32i_kernel - self-attention (S >= 256): a wave owns 32 query rows (v_mfma_f32_32x32x16_bf16), a block of
//     eight waves (or four plus a loader wave) walks the keys in tiles of 64 that arrive by LDS-DMA in a six-slot ring;
//     the softmax instructions are placed in the gaps between the MFMAs of the same wave;
//   * attn16v2_kernel - the 77-token text context (S < 256): a wave owns 16 query rows (v_mfma_f32_16x16x32_bf16),
//     so that the two-tile loop still has enough waves to fill the chip;
//   * both compute the scores transposed, S^T = K Q^T, so a lane holds the scores of ONE query row; the S^T accumulator
//     is reused directly as the B operand of O^T = V^T P^T (no LDS round trip for P); V^T fragments come from
//     ds_read_b64_tr_b16 on the row-major V image; K and V images are XOR-swizzled per 16-byte chunk on the DMA source
//     side so the fragment reads are conflict-free; keys beyond S are zero-filled and masked.
// fp32 kernel ("strict" parity mode): plain FMA online softmax, one query row per thread.
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


__device__ __forceinline__ int swz_k(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_v(int row) { return ((row >> 1) & 1) << 2; }

__device__ __attribute__((aligned(16))) unsigned int g_att_zero16[4] = {0u, 0u, 0u, 0u};
__device__ __forceinline__ int swz_k16(int row) { return row & 7; }
__device__ __forceinline__ int swz_v16(int row) { return ((row >> 1) & 3) << 1; }

// ---- 16-row kernel (v_mfma_f32_16x16x32_bf16), used for the 77-token text context: one wave owns 16 query rows -----
//   S^T[key][q] = K Q^T : A = K rows (lane: key = l&15, d = 32ks + 8g..), B = Q^T (lane: q = l&15, same d);
//                         D: lane (q = l&15, g = l>>4) holds keys 16kb + 4g + r
//   O^T[d][q]  += V^T P^T: B = P^T straight from the S registers of key blocks (2kp, 2kp+1): k-slot 8g + j <-> key
//                         32kp + 16(j>>2) + 4g + (j&3); A = V^T through two transposed 4x16 block reads per fragment
//                         that follow the same key order.
// The softmax is cut to what the VALU cannot avoid.  At D = 64 the kernel is bound by the softmax arithmetic, not by the matrix pipe (per 16 x 64 score tile a wave
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
// K/V tiles of 64 keys arrive by LDS-DMA into a ring of three swizzled buffers, one barrier per tile (the text context
// is two tiles: both are requested in the prologue).
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own pieces of tile kt+1 have landed ...
        __builtin_amdgcn_s_barrier();                              // ... and everyone's; tile kt-1 is dead
        if (kt + 2 < nkt) dma_tile(kt + 2, fb);
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
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][r] = fast_exp2(s[kb][r]);
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

// ---- 32-row kernel, interleaved: softmax instructions placed in the gaps between the MFMAs of the same wave ----------
// At D = 64 a 64-key tile costs a wave 20 MFMAs (640 pipe cycles) and ~80 VALU instructions, 32 of them v_exp_f32 at
// 8 issue cycles: ~420 cycles of vector issue, which fit the 24 free issue cycles of each MFMA gap only if they are
// PLACED there (a wave issues in order: eight MFMAs in a row stall it at the second one, and the VALU behind them
// waits).  One trip of this kernel is a single scheduling region, pinned gap by gap with sched_barrier:
//     QK phase, 8 MFMAs  K(t+1) Q^T -> next scores   | exp + pack of the first half of tile t | V^T(t) fragment reads
//     PV phase, 12 MFMAs V^T(t) P(t) (+ ones block)  | exp + pack of the second half, max of the next scores | K(t+2) reads
// then the counted DMA wait, one block barrier, and the (rare) branch that moves the lazy maximum.  Because P(t) is
// multiplied in the trip that exponentiates it, nothing is pending when the maximum moves.
template <int NW, bool LD>
__global__ __launch_bounds__((NW + (LD ? 1 : 0)) * 64) void attn32i_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                      const bf16* __restrict__ V, bf16* __restrict__ O,
                                                      int T, int S, long ldq, long ldk, long ldv, long ldo, float scale_log2e,
                                                      int H, unsigned long long* probe) {
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
    // 1-D grid, XCD-aware: consecutive block ids go round-robin to the eight XCDs, so XCD c takes the c-th contiguous
    // eighth of the (batch, head, query block) list - the query blocks of one head (they all stream that head's K and V)
    // meet in one L2 instead of eight (HBM-side bytes of the 4096-token launch: 4.5x the algorithmic bytes before)
    int head, b, xblk;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7;
        const int w = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
        const int gx = (T + 32 * NW - 1) / (32 * NW);
        const int hb = w / gx;
        xblk = w - hb * gx;
        b = hb / H;
        head = hb - b * H;
    }
    const int q0 = (xblk * NW + wave) * 32;
    const int qrow = min(q0 + r32, T - 1);

    const bf16* Qb = Q + (size_t)b * T * ldq + (size_t)head * ATT_D;
    const bf16* Kb = K + (size_t)b * S * ldk + (size_t)head * ATT_D;
    const bf16* Vb = V + (size_t)b * S * ldv + (size_t)head * ATT_D;
    const bf16* zeros = reinterpret_cast<const bf16*>(g_att_zero16);

    const int nkt = (S + ATT_KV - 1) / ATT_KV;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(att_lds_offset(lds));
    if constexpr (LD) {
        // ---- loader wave (wave NW): with one wave per SIMD nobody multiplies while a compute wave queues in the load path
        //      (16 wave-instructions of 1 KiB per tile, 16 cycles each), so one extra wave issues every piece of every tile and
        //      the compute waves issue none.  It keeps the compute waves' barrier sequence: three in the prologue (tile 0, 1, 2
        //      landed), then one per trip (tile kt+3 landed, tile kt+4 in flight).
        if (wave == NW) {
            const int lr_ = lane >> 3, pc_ = lane & 7;
            const bf16* src[16];
#pragma unroll
            for (int pce = 0; pce < 16; ++pce) {
                const int isv = pce >> 3, row = (pce & 7) * 8 + lr_;
                const int c = pc_ ^ (isv ? swz_v(row) : swz_k(row));
                src[pce] = isv ? Vb + (size_t)row * ldv + c * 8 : Kb + (size_t)row * ldk + c * 8;
            }
            const long kstep = (long)ATT_KV * ldk, vstep = (long)ATT_KV * ldv;
            int slot_i = 0;
            auto load_tile = [&](int kt) {            // all sixteen pieces of tile kt (< nkt) into the next ring slot
                const unsigned slot = lds0 + slot_i * BUF_B;
                const bool whole = (kt + 1) * ATT_KV <= S;
#pragma unroll
                for (int pce = 0; pce < 16; ++pce) {
                    const int isv = pce >> 3, rb = pce & 7;
                    const bf16* sp = (whole || kt * ATT_KV + rb * 8 + lr_ < S) ? src[pce] : zeros;
                    att_dma16(sp, slot + isv * TILE_B + rb * 1024);
                    src[pce] += isv ? vstep : kstep;
                }
                slot_i = slot_i == RB - 1 ? 0 : slot_i + 1;
            };
            // prologue: tiles 0..2 requested, barrier when tile 0 / 1 / 2 has landed; tile 3 joins after the first barrier
            // (at most 48 DMAs in flight: the vmcnt immediate ends at 63)
            const int n0 = min(nkt, 3);
            for (int kt = 0; kt < n0; ++kt) load_tile(kt);
            if (n0 == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (n0 == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (nkt > 3) { load_tile(3); asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); }      // tile 1 landed (2, 3 fly)
            else if (n0 == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (nkt > 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                          // tile 2 landed (3 flies)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int kt = 0; kt < nkt; ++kt) {
                if (kt + 4 < nkt) { load_tile(kt + 4); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }   // tile kt+3 landed
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            return;
        }
    }

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
    const unsigned dump_off = lds0 + RB * BUF_B;
    auto dma_piece = [&](int kt, int i) {             // piece i of tile kt into ring slot dbuf
        if constexpr (LD) return;                      // (the loader wave issues them)
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
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(qraw[0]), "+v"(qraw[1]), "+v"(qraw[2]), "+v"(qraw[3]) : "n"(LD ? 0 : 3 * PIECES) : "memory");   // Q and tile 0: start on them while the others fly
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
    auto trip = [&](f32x16& s0, f32x16& s1, f32x16& n0, f32x16& n1, int kt) {
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
            const float e0 = fast_exp2(s0[2 * g]), e1 = fast_exp2(s0[2 * g + 1]);
            pb[g >> 2][2 * (g & 3)] = (bf16)e0; pb[g >> 2][2 * (g & 3) + 1] = (bf16)e1;
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
                const float e0 = fast_exp2(s1[2 * p]), e1 = fast_exp2(s1[2 * p + 1]);
                pb[2 + (p >> 2)][2 * (p & 3)] = (bf16)e0; pb[2 + (p >> 2)][2 * (p & 3) + 1] = (bf16)e1;
            }
            if (p >= 2 && p < 6) {                     // one chain start per gap
                const int c = p - 2;
                const f32x16& sx = c < 2 ? n0 : n1;
                const int e = 8 * (c & 1);
                mch[c] = att_max3(sx[e], sx[e + 1], sx[e + 2]);
            }
            if (p >= 6 && p < 8) {
#pragma unroll
                for (int c = 2 * (p - 6); c < 2 * (p - 6) + 2; ++c) {
                    const f32x16& sx = c < 2 ? n0 : n1;
                    const int e = 8 * (c & 1);
                    mch[c] = att_max3(mch[c], sx[e + 3], sx[e + 4]);
                }
            }
            if (p >= 8) {
                const int c = p - 8;
                const f32x16& sx = c < 2 ? n0 : n1;
                const int e = 8 * (c & 1);
                mch[c] = att_max3(mch[c], sx[e + 5], sx[e + 6]);
                mch[c] = att_max3(mch[c], sx[e + 7], sx[e + 7]);
            }
            // LDS-DMA of tile kt+4 (spreading the pieces over the gaps by wave was tried: the per-gap branches cost more
            // than the queueing of sixteen simultaneous wave-instructions in the CU's load path)
            if (p >= 12 - PIECES) dma_piece(kt + 4, p - (12 - PIECES));
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
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");   // own pieces of tile kt+3 have landed; tile kt+4's may fly
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
    if (probe && lane == 0 && q0 == wave * 32 && head == 0 && b == 0) {
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
        static const int force_nw = att_dev_env_int("ST_ATT_NW", 0);       // dev knob: waves per block of the self-attention kernel
        static const int force_16 = att_dev_env_int("ST_ATT_R16", -1);     // dev knob: 1 = 16-row kernel for every S, 0 = never
        const bool rows16 = force_16 >= 0 ? force_16 != 0 : S < 256;
        if (rows16) {
            // measured (tools/op_bench.py): 16-row waves win for the 77-key text context (more waves for a two-tile loop);
            // TAG 1 only gives these launches their own kernel name in the profiles
            hipLaunchKernelGGL((attn16v2_kernel<4, 1>), dim3(cdiv(T, 64), H, B), dim3(256), 0, st, (const bf16*)q, (const bf16*)k,
                               (const bf16*)v, (bf16*)out, T, S, ldq, ldk, ldv, ldo, c);
            return st_check_launch("attention");
        }
        constexpr size_t RING = 6 * 2 * ATT_KV * 128 + 1024;      // six (K, V) tile slots + the dump line of the dummy DMAs
        // eight waves (256 query rows) share a K/V ring unless that leaves half the CUs idle: SDXL's 32x32 level at
        // batch 1 is 80 such blocks; as 160 blocks of four waves every SIMD holds one wave (18 us against 21)
        int nw = (long)cdiv(T, 256) * H * B <= 128 ? 4 : 8;
        if (force_nw == 4 || force_nw == 8) nw = force_nw;
        // four-wave blocks leave one wave per SIMD, with nobody to multiply while a wave queues in the load path: they
        // get a fifth wave that issues every LDS-DMA (16.7 -> 12.4 us with the DMAs compiled out altogether)
        auto kfn = nw == 8 ? attn32i_kernel<8, false> : attn32i_kernel<4, true>;
        static bool once = ((void)hipFuncSetAttribute((const void*)attn32i_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RING),
                            (void)hipFuncSetAttribute((const void*)attn32i_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RING), true);
        (void)once;
        ST_REQUIRE((long)cdiv(T, 32 * nw) * H * B < (1L << 31), "attention: too many blocks");
        hipLaunchKernelGGL(kfn, dim3(cdiv(T, 32 * nw) * H * B), dim3(nw == 8 ? 512 : 320), RING, st, (const bf16*)q, (const bf16*)k, (const bf16*)v,
                           (bf16*)out, T, S, ldq, ldk, ldv, ldo, c, H, ATT_PROBE_ARG);
    } else if (dtype == ST_F32) {
        hipLaunchKernelGGL(attn_f32_kernel, dim3(cdiv(T, 128), H, B), dim3(128), 0, st, (const float*)q, (const float*)k,
                           (const float*)v, (float*)out, T, S, ldq, ldk, ldv, ldo, scale);
    } else {
        return st_fail("attention: unsupported dtype %d", dtype);
    }
    return st_check_launch("attention");
}
