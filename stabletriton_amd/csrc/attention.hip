// Fused attention core for gfx950: out = softmax(q k^T * scale) v, per head, no
// mask, head_dim 64 (every SDXL-base/refiner head).  Row A of SURVEY.md 8a:
// replaces the eager matmul/softmax/matmul of unet_pt.py:133-142 without ever
// materialising the (B,H,T,S) score tensor.
//
// bf16 kernels (flash style):
//   * attn32i_kernel - self-attention (S >= 256): a wave owns 32 query rows (v_mfma_f32_32x32x16_bf16), a block of
//     seven (or four) such waves plus a loader wave walks the keys in tiles of 64 that arrive by LDS-DMA in a six-slot ring;
//     the softmax instructions are placed in the gaps between the MFMAs of the same wave;
//   * attn16v2_kernel - the 77-token text context (S < 256): a wave owns 16 query rows (v_mfma_f32_16x16x32_bf16),
//     so that the two-tile loop still has enough waves to fill the chip;
//   * both compute the scores transposed, S^T = K Q^T, so a lane holds the scores of ONE query row; the S^T accumulator
//     is reused directly as the B operand of O^T = V^T P^T (no LDS round trip for P); V^T fragments come from
//     ds_read_b64_tr_b16 on the row-major V image; K and V images are XOR-swizzled per 16-byte chunk on the DMA source
//     side so the fragment reads are conflict-free; keys beyond S are zero-filled and masked.
// fp32 ("strict" parity mode): attention_f32.hip.
#include "attention_core.h"

template <typename E, int NW, int TAG>
__global__ __launch_bounds__(NW * 64) void attn16v2_kernel(const E* __restrict__ Q, const E* __restrict__ K,
                                                           const E* __restrict__ V, E* __restrict__ O,
                                                           int T, int S, long ldq, long ldk, long ldv, long ldo, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char lds[3 * 2 * ATT_KV * 128];
    const int t_ = threadIdx.x, lane = t_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t_ >> 6);
    const int head = blockIdx.y, b = blockIdx.z;
    const int row0 = blockIdx.x * NW * 16;
    attn16_core<E, NW>(Q + (size_t)b * T * ldq + (size_t)row0 * ldq + (size_t)head * ATT_D, ldq, T - row0,
                    K + (size_t)b * S * ldk + (size_t)head * ATT_D, V + (size_t)b * S * ldv + (size_t)head * ATT_D, ldk, ldv, S,
                    O + (size_t)b * T * ldo + (size_t)row0 * ldo + (size_t)head * ATT_D, ldo, T - row0, scale_log2e, lds, wave, lane);
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
template <typename E, int NW, bool LD>
__global__ __launch_bounds__((NW + (LD ? 1 : 0)) * 64) void attn32i_kernel(const E* __restrict__ Q, const E* __restrict__ K,
                                                      const E* __restrict__ V, E* __restrict__ O,
                                                      int T, int S, long ldq, long ldk, long ldv, long ldo, float scale_log2e,
                                                      int H) {
    typedef typename V16<E>::x8 E8;
    typedef typename V16<E>::x4 E4;
    constexpr int TILE_B = ATT_KV * 128;
    constexpr int BUF_B = 2 * TILE_B;
    constexpr int RB = 6;                             // ring buffers
    constexpr int PIECES = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) char lds[];      // the ring, then a 1-KiB dump for the dummy DMAs

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

    const E* Qb = Q + (size_t)b * T * ldq + (size_t)head * ATT_D;
    const E* Kb = K + (size_t)b * S * ldk + (size_t)head * ATT_D;
    const E* Vb = V + (size_t)b * S * ldv + (size_t)head * ATT_D;
    const E* zeros = reinterpret_cast<const E*>(g_att_zero16);

    const int nkt = (S + ATT_KV - 1) / ATT_KV;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(att_lds_offset(lds));
    if constexpr (LD) {
        // ---- loader wave (wave NW): with one wave per SIMD nobody multiplies while a compute wave queues in the load path
        //      (16 wave-instructions of 1 KiB per tile, 16 cycles each), so one extra wave issues every piece of every tile and
        //      the compute waves issue none.  It keeps the compute waves' barrier sequence: three in the prologue (tile 0, 1, 2
        //      landed), then one per trip (tile kt+3 landed, tile kt+4 in flight).
        if (wave == NW) {
            const int lr_ = lane >> 3, pc_ = lane & 7;
            const E* src[16];
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
                    const E* sp = (whole || kt * ATT_KV + rb * 8 + lr_ < S) ? src[pce] : zeros;
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
    E8 qf[4];
    E8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (E)(r32 == 0 ? 1.0f : 0.0f);

    const int lr = lane >> 3, pc = lane & 7;
    const E* dsrc[PIECES];
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
            const E* src = (live && kt * ATT_KV + rb * 8 + lr < S) ? dsrc[i] : zeros;
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
    E8 kf0[4], kf1[4], vf[4][2], pb[4];
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
        const E8 raw = __builtin_bit_cast(E8, qraw[ks]);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (E)((float)raw[j] * scale_log2e);
    }
    f32x16 sa0, sa1, sb0, sb1;
    {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf0[ks] = *reinterpret_cast<const E8*>(lds + k_off[ks][0]);
            kf1[ks] = *reinterpret_cast<const E8*>(lds + k_off[ks][1]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa0[r] = 0.f; sa1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            sa0 = AttMma<E>::m32(kf0[ks], qf[ks], sa0);
            sa1 = AttMma<E>::m32(kf1[ks], qf[ks], sa1);
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
            kf0[ks] = *reinterpret_cast<const E8*>(lds + BUF_B + k_off[ks][0]);
            kf1[ks] = *reinterpret_cast<const E8*>(lds + BUF_B + k_off[ks][1]);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");           // tile 2: its K fragments are read in the first trip
        __builtin_amdgcn_s_barrier();
    }

    int vslot = 0, kslot = 2;                         // ring slots of tile kt (V^T) and tile kt+2 (K)
    auto trip = [&](f32x16& s0, f32x16& s1, f32x16& n0, f32x16& n1, int kt) {
        const char* vb = lds + vslot * BUF_B + TILE_B;
        const char* kb = lds + kslot * BUF_B;
        float mch[4];                                  // four maximum chains over the next tile's scores
        // ---- QK phase: gap g carries MFMA g, two exponentials + their pack, one V^T fragment (two transposed reads) ----
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int ks = g >> 1;
            if ((g & 1) == 0) n0 = AttMma<E>::m32(kf0[ks], qf[ks], ks == 0 ? negm : n0);
            else              n1 = AttMma<E>::m32(kf1[ks], qf[ks], ks == 0 ? negm : n1);
            const float e0 = fast_exp2(s0[2 * g]), e1 = fast_exp2(s0[2 * g + 1]);
            pb[g >> 2][2 * (g & 3)] = (E)e0; pb[g >> 2][2 * (g & 3) + 1] = (E)e1;
            vf[g >> 1][g & 1] = v_frag<E>(vb, ((g & 1) ? v_base1 : v_base0) + (g >> 1) * 2048, ((g & 1) ? v_base1 : v_base0) + (g >> 1) * 2048 + 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- PV phase: gap p carries MFMA p; gaps 0-7 the other sixteen exponentials, gaps 2-11 the maximum of the next
        //      tile's scores, gaps 4-11 the K fragments of tile kt+2 ----
#pragma unroll
        for (int p = 0; p < 12; ++p) {
            const int s_ = p / 3, w = p % 3;
            if (w == 0)      o0 = AttMma<E>::m32(vf[s_][0], pb[s_], o0);
            else if (w == 1) o1 = AttMma<E>::m32(vf[s_][1], pb[s_], o1);
            else             o2 = AttMma<E>::m32(ones, pb[s_], o2);
            if (p < 8) {
                const float e0 = fast_exp2(s1[2 * p]), e1 = fast_exp2(s1[2 * p + 1]);
                pb[2 + (p >> 2)][2 * (p & 3)] = (E)e0; pb[2 + (p >> 2)][2 * (p & 3) + 1] = (E)e1;
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
                if ((i & 1) == 0) kf0[ks] = *reinterpret_cast<const E8*>(kb + k_off[ks][0]);
                else              kf1[ks] = *reinterpret_cast<const E8*>(kb + k_off[ks][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float mx = att_max3(mch[0], mch[1], att_max3(mch[2], mch[3], mch[3]));
        vslot = vslot == RB - 1 ? 0 : vslot + 1;
        kslot = kslot == RB - 1 ? 0 : kslot + 1;
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
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");   // own pieces of tile kt+3 have landed; tile kt+4's may fly
        dma_next();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        trip(sa0, sa1, sb0, sb1, kt);
        if (kt + 1 < nkt) trip(sb0, sb1, sa0, sa1, kt + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    const float l = __shfl(o2[0], r32, 64);
    const float inv = 1.0f / l;
    if (q0 + r32 < T) {
        E* orow = O + (size_t)b * T * ldo + (size_t)(q0 + r32) * ldo + (size_t)head * ATT_D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            E4 a_, c_;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a_[e] = (E)(o0[4 * g + e] * inv); c_[e] = (E)(o1[4 * g + e] * inv); }
            *reinterpret_cast<E4*>(orow + 8 * g + 4 * h) = a_;
            *reinterpret_cast<E4*>(orow + 32 + 8 * g + 4 * h) = c_;
        }
    }
}

// fp32 ("strict" parity mode): attention_f32.hip - split operands on the 16-bit matrix pipe, fp32 softmax
int attention_f32_launch(const float* q, const float* k, const float* v, float* out, int B, int T, int S, int H,
                         long ldq, long ldk, long ldv, long ldo, float scale, void* out_split, bool presplit, hipStream_t st);

// head_dim 16, 32, 128 (attention_anyd.hip)
int attention_anyd_launch(int dtype, const void* q, const void* k, const void* v, void* out, int B, int T, int S, int H, int D,
                          long ldq, long ldk, long ldv, long ldo, float scale, void* out_split, hipStream_t st);

template <typename E>
static int attention16_launch(const void* q, const void* k, const void* v, void* out, int B, int T, int S, int H,
                              long ldq, long ldk, long ldv, long ldo, float scale, hipStream_t st) {
    const float c = scale * 1.4426950408889634f;
    static const int force_nw = att_dev_env_int("ST_ATT_NW", 0);       // dev knob: waves per block of the self-attention kernel
    static const int force_16 = att_dev_env_int("ST_ATT_R16", -1);     // dev knob: 1 = 16-row kernel for every S, 0 = never
    const bool rows16 = force_16 >= 0 ? force_16 != 0 : S < 256;
    if (rows16) {
        // measured (tools/op_bench.py): 16-row waves win for the 77-key text context (more waves for a two-tile loop);
        // TAG 1 only gives these launches their own kernel name in the profiles
        hipLaunchKernelGGL((attn16v2_kernel<E, 4, 1>), dim3(cdiv(T, 64), H, B), dim3(256), 0, st, (const E*)q, (const E*)k,
                           (const E*)v, (E*)out, T, S, ldq, ldk, ldv, ldo, c);
        return st_check_launch("attention");
    }
    constexpr size_t RING = 6 * 2 * ATT_KV * 128 + 1024;      // six (K, V) tile slots + the dump line of the dummy DMAs
    // Compute waves per block, each with one extra wave that issues every LDS-DMA of every tile (a compute wave that
    // queues in the CU's load path multiplies nothing meanwhile: 16 wave-instructions of 1 KiB per tile, 16 cycles each).
    // Seven compute waves (224 query rows; with the loader two waves per SIMD) share a K/V ring unless that leaves half
    // the CUs idle: SDXL's 32x32 level at batch 1 is then 160 blocks of four waves, one per SIMD (15.4 us against 20.2).
    // Against eight self-loading waves: 4096 tokens 73 -> 70 us (batch 1), 215 -> 206 us (batch 4); 1024 tokens at
    // batch 4 43.8 -> 39.2 us.
    // Three compute waves (96 rows) when four would still leave a third of the CUs without a block: the 32x32 level at batch 1
    // is 220 blocks instead of 160 (15.3 -> 13.9 us).
    int nw = (long)cdiv(T, 256) * H * B <= 128 ? 4 : 7;
    if (nw == 4 && (long)cdiv(T, 128) * H * B <= 176 && (long)cdiv(T, 96) * H * B <= 256) nw = 3;
    // Eight self-loading waves (256 rows per block) where that saves a whole round of blocks over seven + loader (224 rows): a round
    // of the eight-wave blocks takes 1.2x as long (tools/attn_nw_sweep.py: the refiner's 12 heads x 4,096 tokens at batch 4 are 878
    // blocks = 4 rounds of 65 us against 768 = 3 rounds of 77 us)
    if (nw == 7) {
        const long r7 = cdiv((long)cdiv(T, 224) * H * B, 256L), r8 = cdiv((long)cdiv(T, 256) * H * B, 256L);
        if (6 * r8 < 5 * r7) nw = 8;
    }
    if (force_nw == 3 || force_nw == 4 || force_nw == 8 || force_nw == 7) nw = force_nw;
    auto kfn = nw == 8 ? attn32i_kernel<E, 8, false> : nw == 7 ? attn32i_kernel<E, 7, true> : nw == 3 ? attn32i_kernel<E, 3, true> : attn32i_kernel<E, 4, true>;
    int threads = nw == 4 ? 320 : nw == 3 ? 256 : 512;
#ifdef ST_DEV_CONFIGS
    if (force_nw == 5) { nw = 5; kfn = attn32i_kernel<E, 5, true>; threads = 384; }
    if (force_nw == 6) { nw = 6; kfn = attn32i_kernel<E, 6, true>; threads = 448; }
#endif
    static unsigned long long lds_ok[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // per kernel: bit mask over device ordinals
    ensure_dynamic_lds(kfn, RING, &lds_ok[nw - 1]);
    ST_REQUIRE((long)cdiv(T, 32 * nw) * H * B < (1L << 31), "attention: too many blocks");
    hipLaunchKernelGGL(kfn, dim3(cdiv(T, 32 * nw) * H * B), dim3(threads), RING, st, (const E*)q, (const E*)k, (const E*)v,
                       (E*)out, T, S, ldq, ldk, ldv, ldo, c, H);
    return st_check_launch("attention");
}

extern "C" int st_attention(const void* q, const void* k, const void* v, void* out, int B, int T, int S, int H, int D,
                            long ldq, long ldk, long ldv, long ldo, float scale, int dtype, void* stream) {
    ST_REQUIRE(q && k && v && out, "attention: null pointer");
    ST_REQUIRE(B > 0 && T > 0 && S > 0 && H > 0, "attention: bad shape B=%d T=%d S=%d H=%d", B, T, S, H);
    ST_REQUIRE(D == 16 || D == 32 || D == 64 || D == 128, "attention: head_dim %d not supported (16, 32, 64, 128)", D);
    ST_REQUIRE(H <= 65535 && B <= 65535, "attention: too many heads/batches for one launch");
    const int vec = st_dtype_is16(dtype) ? 8 : 4;
    ST_REQUIRE(ldq % vec == 0 && ldk % vec == 0 && ldv % vec == 0 && ldo % 4 == 0, "attention: strides must keep 16-byte alignment");
    ST_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0, "attention: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    void* out_split = nullptr;
    if (int e = st_take_split_arm("attention", (long)B * T, H * D, dtype == ST_F32, &out_split)) return e;
#ifdef ST_DEV_CONFIGS
    static const int force_anyd = att_dev_env_int("ST_ATT_ANYD", 0);       // dev knob: head_dim 64 on the generic kernel too
    if (force_anyd) return attention_anyd_launch(dtype, q, k, v, out, B, T, S, H, D, ldq, ldk, ldv, ldo, scale, out_split, st);
#endif
    if (D != ATT_D) return attention_anyd_launch(dtype, q, k, v, out, B, T, S, H, D, ldq, ldk, ldv, ldo, scale, out_split, st);
    if (dtype == ST_BF16) return attention16_launch<bf16>(q, k, v, out, B, T, S, H, ldq, ldk, ldv, ldo, scale, st);
    if (dtype == ST_F16) return attention16_launch<f16>(q, k, v, out, B, T, S, H, ldq, ldk, ldv, ldo, scale, st);
    if (dtype == ST_F32) return attention_f32_launch((const float*)q, (const float*)k, (const float*)v, (float*)out, B, T, S, H, ldq, ldk, ldv, ldo, scale, out_split, false, st);
    return st_fail("attention: unsupported dtype %d", dtype);
}

// st_attention for fp32 tensors whose K and V a producer left as split images (the q|k|v projection of self-attention:
// st_arm_split_output): ks / vs = first byte of head 0's columns in row 0 of the image(s), k_cols / v_cols = values per image
// row; q, out and everything else as st_attention with ST_F32.  Same arithmetic as st_attention (which splits K / V itself,
// tile by tile): the same bits.
extern "C" int st_attention_split(const void* q, const void* ks, const void* vs, void* out, int B, int T, int S, int H, int D,
                                  long ldq, long k_cols, long v_cols, long ldo, float scale, void* stream) {
    ST_REQUIRE(q && ks && vs && out, "attention_split: null pointer");
    ST_REQUIRE(B > 0 && T > 0 && S > 0 && H > 0, "attention_split: bad shape B=%d T=%d S=%d H=%d", B, T, S, H);
    ST_REQUIRE(D == ATT_D, "attention_split: head_dim %d not supported (only %d)", D, ATT_D);
    ST_REQUIRE(H <= 65535 && B <= 65535, "attention_split: too many heads/batches for one launch");
    ST_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && k_cols % 32 == 0 && v_cols % 32 == 0, "attention_split: strides must keep 16-byte alignment, image rows whole segments");
    ST_REQUIRE(((uintptr_t)q | (uintptr_t)out) % 16 == 0 && ((uintptr_t)ks | (uintptr_t)vs) % 128 == 0, "attention_split: q / out 16-byte, image columns segment (128-byte) aligned");
    void* out_split = nullptr;
    if (int e = st_take_split_arm("attention_split", (long)B * T, H * D, true, &out_split)) return e;
    return attention_f32_launch((const float*)q, (const float*)ks, (const float*)vs, (float*)out, B, T, S, H, ldq, k_cols, v_cols, ldo, scale, out_split, true,
                                (hipStream_t)stream);
}
