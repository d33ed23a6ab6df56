// Shared device code of the attention kernels (attention.hip) and of the GEMM that fuses the text-context attention into
// the query projection (gemm.hip).  Internal to csrc/.
#pragma once
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

static __device__ __attribute__((aligned(16))) unsigned int g_att_zero16[4] = {0u, 0u, 0u, 0u};
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
// (the 16-bit-integer form of the builtin: the read moves bits, so one form serves bf16 and f16)
typedef __attribute__((ext_vector_type(4))) short att_s16x4;
typedef __attribute__((ext_vector_type(8))) short att_s16x8;
typedef __attribute__((address_space(3))) att_s16x4 att_lds_s16x4;
template <typename E>
__device__ __forceinline__ typename V16<E>::x8 v_frag(const char* lds_base, int off_lo, int off_hi) {
    const att_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4*)(lds_base + off_lo));
    const att_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4*)(lds_base + off_hi));
    return __builtin_bit_cast(typename V16<E>::x8, (att_s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
// eight probabilities -> one MFMA operand: four packed conversions (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
template <typename E>
__device__ __forceinline__ typename V16<E>::x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    typename V16<E>::x8 r;
    r[0] = (E)a0; r[1] = (E)a1; r[2] = (E)a2; r[3] = (E)a3; r[4] = (E)a4; r[5] = (E)a5; r[6] = (E)a6; r[7] = (E)a7;
    return r;
}
// the two MFMA shapes of the attention kernels, per 16-bit element type (same cycles for bf16 and f16)
template <typename E> struct AttMma;
template <> struct AttMma<bf16> {
    static __device__ __forceinline__ f32x4 m16(const bf16x8& a, const bf16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 m32(const bf16x8& a, const bf16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct AttMma<f16> {
    static __device__ __forceinline__ f32x4 m16(const f16x8& a, const f16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 m32(const f16x8& a, const f16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// Three-way maximum, deliberately NOT inline asm: the scores it reads come straight out of MFMAs, and the hardware does
// not interlock an MFMA result against a VALU read - the compiler inserts the wait states, but only for instructions it
// can see.  An asm v_max3_f32 here read accumulators that were still being written whenever the matrix pipe was shared
// with another kernel (tools/att_race.py: output changed by 1 ulp when a second stream kept the CUs busy).
__device__ __forceinline__ float att_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// The body of the 16-row kernel as a device function, so that the GEMM that projects the queries of the text-context
// attention can run it as its epilogue on the Q tile it has just left in LDS (gemm.hip, st_ln_linear_xattn).
//   Qb / Ob: row 0 of this block's NW*16 queries / outputs of ONE head (global or LDS for Qb); rows >= q_rows are clamped,
//   rows >= o_rows not stored; Kb / Vb: key 0 of that head; lds: 3 * 16 KiB ring, 16-byte aligned.
template <typename E, int NW>
__device__ __forceinline__ void attn16_core(const E* Qb, long ldq, int q_rows, const E* Kb, const E* Vb, long ldk, long ldv,
                                            int S, E* Ob, long ldo, int o_rows, float scale_log2e, char* lds, int wave, int lane) {
    typedef typename V16<E>::x8 E8;
    typedef typename V16<E>::x4 E4;
    constexpr int TILE_B = ATT_KV * 128;
    constexpr int BUF_B = 2 * TILE_B;
    constexpr int PIECES = 16 / NW;
    static_assert(NW <= 16 && 16 % NW == 0, "waves must divide the 16 DMA pieces of a tile");
    const int c16 = lane & 15, g = lane >> 4;
    const int q0 = wave * 16;
    const int qrow = min(q0 + c16, q_rows - 1);
    const E* zeros = reinterpret_cast<const E*>(g_att_zero16);

    E8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const E8 raw = *reinterpret_cast<const E8*>(Qb + (size_t)qrow * ldq + 32 * ks + 8 * g);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (E)((float)raw[j] * scale_log2e);
    }
    // V^T "row 64": ones for the lanes that hold d = 0 of the extra block, zeros elsewhere
    E8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (E)(c16 == 0 ? 1.0f : 0.0f);

    // LDS-DMA sources: one running pointer per piece, advanced by 64 keys per tile (no per-tile address arithmetic);
    // only a tile that reaches past S takes the checked form (rows beyond S read a zero line)
    const int lr = lane >> 3, pc = lane & 7;
    const E* dsrc[PIECES];
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
            const E* src = dsrc[i];
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
            const E8 ka = *reinterpret_cast<const E8*>(kb_ + kb * 2048 + k_off0);
            const E8 kc = *reinterpret_cast<const E8*>(kb_ + kb * 2048 + k_off1);
            f32x4 acc = {-m_ref, -m_ref, -m_ref, -m_ref};
            acc = AttMma<E>::m16(ka, qf[0], acc);
            s[kb] = AttMma<E>::m16(kc, qf[1], acc);
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
        E8 vf[2][4];
        {
            const char* vb = lds + cur * BUF_B + TILE_B;
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                vf[kp][0] = v_frag<E>(vb, v_off0 + kp * 4096, v_off0 + kp * 4096 + 2048);
                vf[kp][1] = v_frag<E>(vb, v_off1 + kp * 4096, v_off1 + kp * 4096 + 2048);
                vf[kp][2] = v_frag<E>(vb, v_off2 + kp * 4096, v_off2 + kp * 4096 + 2048);
                vf[kp][3] = v_frag<E>(vb, v_off3 + kp * 4096, v_off3 + kp * 4096 + 2048);
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
            const E8 pb = pack8<E>(s[2 * kp][0], s[2 * kp][1], s[2 * kp][2], s[2 * kp][3],
                                    s[2 * kp + 1][0], s[2 * kp + 1][1], s[2 * kp + 1][2], s[2 * kp + 1][3]);
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db] = AttMma<E>::m16(vf[kp][db], pb, o[db]);
            o[4] = AttMma<E>::m16(ones, pb, o[4]);
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
    if (q0 + c16 < o_rows) {
        E* orow = Ob + (size_t)(q0 + c16) * ldo;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            E4 a_;
#pragma unroll
            for (int e = 0; e < 4; ++e) a_[e] = (E)(o[db][e] * inv);
            *reinterpret_cast<E4*>(orow + 16 * db + 4 * g) = a_;
        }
    }
}

