// GEMM-shaped operators: the eight-phase 256-row kernel (large Linear problems; 16-bit, e4m3 and split fp32 operands) and its launchers.
// Internal to csrc/.
#pragma once
#include "gemm_dma.h"

// =============================================================================
// gemm8p: 256-row tiles for the large Linear problems (the projections at batch >= 2, the GEGLU projection at any
// batch).  A 128 x 128 tile needs (128+128)*128 B of LDS fill per 0.21 us of MFMA work - more than the ~130 GB/s one
// CU pulls from its L2 - a 256-row tile about half of that.  Two shapes of the same kernel:
//     256 x 256: 8 waves = 2 (rows) x 4 (columns), wave tile 128 x 64;
//     256 x 160: 8 waves = 4 x 2, wave tile 64 x 80 - SDXL's widths are 5 * 2^k: 1024 x 10240 (the GEGLU projection at
//                batch 1) is 160 tiles of 256 x 256 but 256 of 256 x 160, one per CU.
//   * A K tile is two phases per wave group (round 5; four quadrant phases before): (A0, B) then (A1, B), Ah = the two halves
//     of the wave's rows, B = all its accumulator columns.  The first phase reads A0 and B, the second only A1.
//   * The two halves of the block's waves run half a phase apart (waves 4-7 pass one extra barrier first): while one
//     half multiplies, the other reads fragments and issues DMAs, on the same SIMDs - a software ping-pong with two raw
//     barriers per phase and no wave ever doing both at once.
//   * LDS: two K tiles, each as four regions (A0, A1, B0, B1), filled by LDS-DMA (two 1-KiB pieces per wave and region; a
//     region with fewer than sixteen pieces fills up with dummy pieces so that every wave counts the same vmcnt), swizzled on
//     the source side as in gemm_dma_kernel.  A region is refilled in the first half phase after both wave groups have read
//     it and waited for (counted vmcnt, never 0) four half phases later.
// Epilogue: straight from the accumulator registers over a permuted staging of W where the launch's feature set has such an
// instance (DIRECT; epilogue.h, direct epilogue), else staged through LDS (GEGLU: tile columns [values | gates]).  LayerNorm
// folding, statistics, next-weights touches and the XCD-aware tile order are the ones of gemm_dma_kernel.  No K split.
// =============================================================================
template <int BN, int WGM, int WGN> constexpr bool TILE_OK_SPLIT() { return (256 / WGM / 16) * (BN / WGN / 16) <= 20; }
template <typename T, bool GEGLU, bool LNF, int BN = 256, int WGM = 2, int WGN = 4, bool DIRECT = false>
__global__ __launch_bounds__(512) void gemm8p_kernel(const GemmArgs p) {
    static_assert(sizeof(T) <= 2 || is_split<T>(), "16-bit elements (bf16 / f16), e4m3 bytes, or split fp32 operands (strict mode: 256 x 160 tiles)");
    static_assert(!is_split<T>() || (TILE_OK_SPLIT<BN, WGM, WGN>()), "split operands carry two accumulator sets: wave tiles of at most 64 x 80");
    constexpr int BM = 256, NW = 8;
    constexpr int KB = 128 / (int)sizeof(T);           // elements per 128-byte row of a K tile: 64, 128 e4m3, or 32 split fp32 (hi chunks 0-3, lo chunks 4-7)
    constexpr int EV = 16 / (int)sizeof(T);            // elements per 16-byte chunk
    typedef typename OutT<T>::type TO;                // element type of C, bias, residual (e4m3 operands: bf16)
    static_assert(WGM * WGN == NW && BM % (32 * WGM) == 0 && BN % (16 * WGN) == 0 && (!GEGLU || BN % 32 == 0), "wave layout");
    constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 16, TN = WTN / 16;
    constexpr int TMH = TM / 2, TN0 = (TN + 1) / 2, TN1 = TN - TN0;          // accumulator tiles per A half / in B0 / in B1
    constexpr int RA = WGM * TMH * 16, RB0 = WGN * TN0 * 16, RB1 = WGN * TN1 * 16;      // rows of the regions
    static_assert(RA == 128 && RB0 <= 128 && RB1 <= 128 && RB0 % 8 == 0 && RB1 % 8 == 0, "a region is at most sixteen 8-row pieces");
    constexpr int HA = RA * 128, HB0 = RB0 * 128, HB1 = RB1 * 128;            // bytes
    constexpr int TILE_B = 2 * HA + HB0 + HB1;                                // A0 A1 B0 B1
    constexpr int BNO = GEGLU ? BN / 2 : BN;
    typedef typename Mma<T>::Frag Frag;
    // what one ds_read_b128 delivers: a whole MFMA operand of 32 k (16-bit), or half of the 128-k operand of the e4m3 instruction
    typedef typename std::conditional<frag2<T>(), u32x4, Frag>::type Half;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const lnrows = lds + 2 * TILE_B;            // LayerNorm (mean, rstd) per row
    char* const dump = lnrows + BM * 8;               // target of the dummy DMAs

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int tiles_m = p.M / BM;
    const int nblk = gridDim.x - p.helper_blocks, bid = blockIdx.x;
    if (bid >= nblk) {                               // helper block on an otherwise idle CU: the next launch's weights
        unsigned int sink = 0;
        touch_next_weights(p, sink, true);
        retire_touches(sink);
        return;
    }
    const TileId tid = tile_of_block(p, bid, nblk);
    const int tile_m = tid.tile_m, tile_n = tid.tile_n;
    const int m0 = tile_m * BM, n0 = tile_n * BNO;
#ifdef ST_PROBE8      // developer build (tools/gemm8p_probe.py): cycle stamps of wave 0 into the split-K workspace, 32 bytes per block
    unsigned long long pr_t0 = __builtin_readcyclecounter(), pr_t1 = 0, pr_t2 = 0, pr_r0 = __builtin_amdgcn_s_memrealtime();      // (realtime: constant 100 MHz)
#endif
    const T* __restrict__ Ap = (const T*)p.A;
    const T* __restrict__ Wp = (const T*)p.W;
    const T* zeros = reinterpret_cast<const T*>(g_zero16);

    // ---- per-lane DMA sources: piece e (0, 1) of this wave inside a region covers region rows idx = (2*wave+e)*8 + lr.
    //      A region row idx = wave row (idx / (TMH*16)), row inside that wave's half (idx % (TMH*16)); B likewise with the
    //      wave column.  Tile column c -> row of W: c (plain), or value row c / gate row N + c - BN/2 (GEGLU).
    const int lr = lane >> 3;
    const int lc = (lane & 7) ^ lr;                  // logical 16-byte chunk this lane fetches (source-side swizzle)
    const T* a_src[2];
    const T* b0_src[2];
    const T* b1_src[2];
    auto w_row = [&](int c) { return GEGLU ? (c < BN / 2 ? (size_t)(n0 + c) : (size_t)p.Ng + n0 + (c - BN / 2)) : (size_t)(n0 + c); };
    // DIRECT: W staged permuted (epilogue.h, direct epilogue): accumulator tile j of wave column wc, tile row rho -> row of W
    auto w_row_direct = [&](int wc, int j, int rho) {
        const DirectCol dc = direct_col<TN, GEGLU>(j, rho);
        return (size_t)(dc.gate ? p.Ng : 0) + n0 + wc * (GEGLU ? WTN / 2 : WTN) + dc.col;
    };
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = (2 * wave + e) * 8 + lr;
        a_src[e] = Ap + (size_t)(m0 + (idx / (TMH * 16)) * WTM + (idx % (TMH * 16))) * p.lda + lc * EV;      // half h adds TMH*16 rows
        const int i0 = idx < RB0 ? idx : 0, i1 = idx < RB1 ? idx : 0;
        if constexpr (DIRECT) {
            const int in0 = i0 % (TN0 * 16), in1 = i1 % (TN1 * 16);
            b0_src[e] = Wp + w_row_direct(i0 / (TN0 * 16), in0 >> 4, in0 & 15) * p.K + lc * EV;
            b1_src[e] = Wp + w_row_direct(i1 / (TN1 * 16), TN0 + (in1 >> 4), in1 & 15) * p.K + lc * EV;
        } else {
            b0_src[e] = Wp + w_row((i0 / (TN0 * 16)) * WTN + (i0 % (TN0 * 16))) * p.K + lc * EV;
            b1_src[e] = Wp + w_row((i1 / (TN1 * 16)) * WTN + TN0 * 16 + (i1 % (TN1 * 16))) * p.K + lc * EV;
        }
    }
    const size_t a_half = (size_t)(TMH * 16) * p.lda;
    const int nk = p.K / KB;

    // region r of a K tile: 0 = A0, 1 = A1, 2 = B0, 3 = B1.  Always two DMAs per wave: pieces beyond the region's rows and
    // `kt >= nk` are dummies (a zero line into the dump area).
    auto issue_half = [&](int kt, int region) {
#ifdef ST_PROBE8
        if (p.korder & 2) kt = nk;                    // (developer probe: only dummy DMAs - one zero line each: the loop without its fill traffic)
#endif
        const int roff = region == 0 ? 0 : region == 1 ? HA : region == 2 ? 2 * HA : 2 * HA + HB0;
        const int rrows = region < 2 ? RA : region == 2 ? RB0 : RB1;
        const unsigned dst = lds_addr_of(lds) + (kt & 1) * TILE_B + roff + (2 * wave) * 1024;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool live = kt < nk && (2 * wave + e) * 8 < rrows;
            const T* src = region < 2 ? a_src[e] + (region & 1) * a_half : region == 2 ? b0_src[e] : b1_src[e];
            src = live ? src + (size_t)kt * KB : zeros;
            dma16_at<0>(src, live ? dst + e * 1024 : lds_addr_of(lds) + 2 * TILE_B + BM * 8);        // (= dump)
        }
    };

    // touch the epilogue's operands now (they are first read after the K loop, where a miss would be exposed)
    unsigned int touch_sink = 0;
    {
        auto touch_at = [&](const char* a) {
            a = (const char*)((uintptr_t)a & ~(uintptr_t)3);
            asm volatile("global_load_dword %0, %1, off" : "+v"(touch_sink) : "v"(a) : "memory");
        };
        auto touch = [&](const void* base, long byte_off, int nbytes) {
            for (int o = t * 128; o < nbytes; o += 512 * 128) touch_at((const char*)base + byte_off + o);
        };
        if (p.epi & ST_EPI_BIAS) {
            touch(p.bias, (long)n0 * (long)sizeof(TO), BNO * (int)sizeof(TO));
            if (GEGLU) touch(p.bias, ((long)p.Ng + n0) * (long)sizeof(TO), BNO * (int)sizeof(TO));
        }
        if (LNF) {
            touch(p.ln_c, (long)n0 * 4, BNO * 4); touch(p.ln_d, (long)n0 * 4, BNO * 4);
            if (GEGLU) { touch(p.ln_c, ((long)p.Ng + n0) * 4, BNO * 4); touch(p.ln_d, ((long)p.Ng + n0) * 4, BNO * 4); }
        }
        if (p.epi & ST_EPI_RESIDUAL) {
            constexpr int lines = (BNO * (int)sizeof(TO) + 127) / 128;
            for (int o = t; o < BM * lines; o += 512) {
                const int r = o / lines, l = o - r * lines;
                touch_at((const char*)p.residual + ((size_t)(m0 + r) * p.ldr + n0) * sizeof(TO) + l * 128);
            }
        }
    }
    // LayerNorm-folded GEMM: row statistics from the producer's partials (two threads per row)
    LnRowSum<2> ln_sum;
    if constexpr (LNF) {
        const float2* st2 = reinterpret_cast<const float2*>(p.ln_stats);
        const int row = t >> 1, part = t & 1;
        ln_sum.load(st2 + (size_t)(m0 + row) * p.ln_chunks, p.ln_chunks, part);
    }
    // prologue: K tile 0 whole and A0, B0, B1 of K tile 1 (the loop issues A1(1), then A0, B0, B1 of tile 2, A1(2), ...)
    issue_half(0, 0); issue_half(0, 2); issue_half(0, 3); issue_half(0, 1); issue_half(1, 0); issue_half(1, 2); issue_half(1, 3);
    if constexpr (LNF) {
        const float2* st2 = reinterpret_cast<const float2*>(p.ln_stats);
        const int row = t >> 1, part = t & 1;
        float a1, a2;
        ln_sum.finish(st2 + (size_t)(m0 + row) * p.ln_chunks, p.ln_chunks, part, a1, a2);
        const float mean = a1 / (float)p.K;
        const float rstd = rsqrtf(fmaxf(a2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
        if (part == 0) reinterpret_cast<float2*>(lnrows)[row] = make_float2(mean, rstd);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    wait_vmcnt<8>();                                 // A0(0), B0(0), B1(0) have landed (fourteen DMAs issued; and every load older than the DMAs)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::"v"(touch_sink));
    if (wave >= 4) __builtin_amdgcn_s_barrier();     // the second half of the waves runs one barrier (half a phase) behind the first

    f32x4 acc[TM][TN];
    f32x4 corr[is_split<T>() ? TM : 1][is_split<T>() ? TN : 1];      // split operands: the cross products, in units of 2^-11 (gemm_dma_kernel)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (is_split<T>()) corr[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    const int r16 = lane & 15, q = lane >> 4;
    // fragment addresses: region row = w * (tiles * 16) + frag * 16 + r16, chunk 4*kk + q, swizzled by row & 7 = r16 & 7
    int a_off[2], b0_off[2], b1_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((4 * kk + q) ^ (r16 & 7)) << 4;
        a_off[kk] = (wm * TMH * 16 + r16) * 128 + sw;
        b0_off[kk] = 2 * HA + (wn * TN0 * 16 + r16) * 128 + sw;
        b1_off[kk] = 2 * HA + HB0 + (wn * TN1 * 16 + r16) * 128 + sw;
    }
    // A half in use and the wave's whole B.  16-bit: [..][kk] = the operand of k step kk; e4m3: [..][0] is the whole
    // 128-k operand, assembled from the two 16-byte reads (chunks q and q + 4) where they land; split fp32: the same two reads
    // are the hi and the lo halves of one 32-k operand
    constexpr int NKK = frag2<T>() ? 1 : 2;
    Frag fa[TMH][NKK], fb[TN][NKK];
    auto read_op = [&](const char* base, const int (&off)[2], Frag (&dst)[NKK]) {
        if constexpr (frag2<T>()) {
            const Half lo = *reinterpret_cast<const Half*>(base + off[0]), hi = *reinterpret_cast<const Half*>(base + off[1]);
            dst[0] = Frag{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) dst[kk] = *reinterpret_cast<const Frag*>(base + off[kk]);
        }
    };
    auto read_a = [&](const char* tile, int h) {
#pragma unroll
        for (int i = 0; i < TMH; ++i) read_op(tile + h * HA + i * 2048, a_off, fa[i]);
    };
    auto read_b = [&](const char* tile) {
#pragma unroll
        for (int j = 0; j < TN0; ++j) read_op(tile + j * 2048, b0_off, fb[j]);
#pragma unroll
        for (int j = 0; j < TN1; ++j) read_op(tile + j * 2048, b1_off, fb[TN0 + j]);
    };
    auto half_tile = [&](auto mh_) {                  // the MFMAs of one row half of the wave tile: TMH x TN accumulator tiles, all of the K tile
        constexpr int mh = decltype(mh_)::value;
#ifdef ST_PROBE8
        if (p.korder & 1) return;                     // (developer probe: the loop without its MFMAs - the load side's own pace)
#endif
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
            for (int i = 0; i < TMH; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (is_split<T>()) Mma<T>::run2(acc[mh * TMH + i][j], corr[mh * TMH + i][j], fb[j][kk], fa[i][kk]);
                    else Mma<T>::run(acc[mh * TMH + i][j], fb[j][kk], fa[i][kk]);
                }
        __builtin_amdgcn_s_setprio(0);
    };
    // Round 5: a K tile is TWO phases per wave group, not four - (A0, B) then (A1, B): half as many barriers and counted waits
    // per K tile (each of the eight cost ~75 cycles beside its 256 of MFMA: 2,650-2,850 cycles per K tile against 2,048; the
    // fragments in flight are the same 64 registers: one A half + the wave's whole B instead of one A half + B0 + B1).
    //   L0: read A0, B (12-14 ds_read_b128 pairs)  | issue A1(kt+1)                  | vmcnt(8), barrier | M0: TMH x TN x NKK MFMAs | barrier
    //   L1: read A1                                 | issue A0, B0, B1 of tile kt+2  | vmcnt(8), barrier | M1                        | barrier
    // Every region is requested as soon as both wave groups have read its previous content (A1's buffer after L1 of the tile
    // before, the others after L0 of this tile) and waited for four half phases later: after the issue of a half phase the eight
    // youngest DMAs (this half's and the one before) may stay in flight.
#define ST_PHASE_SYNC()                                   \
    __builtin_amdgcn_sched_barrier(0);                    \
    wait_vmcnt<8>();                                      \
    __builtin_amdgcn_s_barrier();                         \
    __builtin_amdgcn_sched_barrier(0)
#define ST_PHASE_END()                                    \
    __builtin_amdgcn_sched_barrier(0);                    \
    __builtin_amdgcn_s_barrier();                         \
    __builtin_amdgcn_sched_barrier(0)
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
#ifdef ST_PROBE8
    pr_t1 = __builtin_readcyclecounter();
#endif
    for (int kt = 0; kt < nk; ++kt) {
        const char* tile = lds + (kt & 1) * TILE_B;
        // L0 / M0: (A0, B); A1 of tile kt+1 (its buffer was last read in L1 of tile kt-1)
        read_a(tile, 0); read_b(tile);
        issue_half(kt + 1, 1);
        ST_PHASE_SYNC();
        half_tile(I0{});
        ST_PHASE_END();
        // L1 / M1: (A1, B from registers); A0, B0, B1 of tile kt+2 (their buffers were last read in L0 of this tile)
        read_a(tile, 1);
        issue_half(kt + 2, 0); issue_half(kt + 2, 2); issue_half(kt + 2, 3);
        ST_PHASE_SYNC();
        half_tile(I1{});
        ST_PHASE_END();
    }
#undef ST_PHASE_SYNC
#undef ST_PHASE_END
    if (wave < 4) __builtin_amdgcn_s_barrier();      // barrier counts of the two halves are equal again
    wait_vmcnt<0>();                                  // no LDS-DMA may outlive the workgroup's LDS allocation
    __builtin_amdgcn_s_barrier();
#ifdef ST_PROBE8
    pr_t2 = __builtin_readcyclecounter();
#endif
    if constexpr (is_split<T>()) {                    // one fused multiply-add per element: the same bits whatever follows (gemm_dma_kernel)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = __builtin_fmaf(corr[i][j][e], ST_SPLIT_INV, acc[i][j][e]);
    }
    if constexpr (DIRECT)
        direct_epilogue<TO, TM, TN, WTM, WTN, WGM, WGN, GEGLU, LNF>(p, acc, m0, n0, tile_n, wm, wn, r16, q, lds, reinterpret_cast<const float2*>(lnrows),
                                                                    lds_addr_of(lds) + 2 * TILE_B + BM * 8);
    else
        staged_epilogue<TO, BM, BN, WGM, WGN, TM, TN, GEGLU, 2 * TILE_B, !LNF, is_fp8<T>()>(p, acc, m0, n0, tile_n, wm, r16, q, ColsPlain{wn, WTN}, lds,
                                                                            reinterpret_cast<const float2*>(lnrows));
#ifdef ST_PROBE8
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores acknowledged: the block's whole life
    if (p.partial && t == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.partial) + 65536) + (size_t)bid * 8;
        o[0] = pr_t0; o[1] = pr_t1; o[2] = pr_t2; o[3] = __builtin_readcyclecounter(); o[4] = pr_r0; o[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// the two shapes of gemm8p: 256 (2 x 4 waves) and 160 columns (4 x 2 waves)
static inline bool gemm8p_applies(const GemmArgs& a, int bn, int kb = 64) {
    const long n_rows = (a.epi & ST_EPI_GEGLU) ? 2L * a.N : a.N;
    return a.M % 256 == 0 && n_rows % bn == 0 && a.K % kb == 0 && a.K >= 2 * kb && a.N % 8 == 0 &&
           !(a.epi & ST_EPI_ROWBIAS) && (!a.row_stats || !(a.epi & ST_EPI_GEGLU));
}

template <typename T, bool GEGLU, bool LNF, int BN, int WGM, int WGN, bool DIRECT = false>
static void gemm8p_go(const GemmArgs& a, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)(2 * 128 * 128 + BN * 128) + 256 * 8 + 1024;
    auto kfn = gemm8p_kernel<T, GEGLU, LNF, BN, WGM, WGN, DIRECT>;
    static unsigned long long lds_ok = 0;
    ensure_dynamic_lds(kfn, lds, &lds_ok);
    GemmArgs b = a;
    const int tiles_m = a.M / 256, tiles_n = (int)(((a.epi & ST_EPI_GEGLU) ? 2L * a.N : a.N) / BN);
    {   // XCD partition of the tile order: bytes from beyond L2 ~ A * (8 / panels) + W * panels
        const double abytes = (double)a.M * a.K, wbytes = (double)tiles_n * BN * a.K;
        int best_p = 1;
        double best = 1e300;
        for (int pm = 1; pm <= 8 && pm <= tiles_m; pm *= 2) {
            const double c = abytes * (8.0 / pm) + wbytes * pm;
            if (c < best) { best = c; best_p = pm; }
        }
        b.panel_h = cdiv(tiles_m, best_p);
    }
    const int main_blocks = tiles_m * tiles_n;
    b.splitk = 1;
    fill_tile_map(b, tiles_m, tiles_n, 0);
    b.helper_blocks = (b.next_w && main_blocks <= 208) ? (256 - main_blocks > 96 ? 96 : 256 - main_blocks) : 0;
    b.stats_chunks = tiles_n;
    if (a.stats_chunks_out) *a.stats_chunks_out = a.row_stats ? b.stats_chunks : 0;
    if (!colstats_ok(a, 256, LNF)) b.col_stats = nullptr;
    fill_next_per(b, main_blocks + b.helper_blocks);
    hipLaunchKernelGGL(kfn, dim3(main_blocks + b.helper_blocks), dim3(512), lds, st, b);
}

// Does the launch take the direct (register) epilogue?  Its feature set has an instance (epilogue.h: direct_flags) and every
// 16-byte access is aligned; 16-bit outputs only (the e4m3 instances keep the staged form: scales, e4m3 copies).
template <typename T>
static inline bool gemm8p_direct(const GemmArgs& a, bool tall) {
    if constexpr (sizeof(T) != 2) return false;
    static const bool off = dev_env_int("ST_8P_STAGED", 0) != 0;      // (developer A/B)
    if (off || direct_flags(a, a.ln_c != nullptr, tall) < 0) return false;
    auto al16 = [](const void* p_) { return ((uintptr_t)p_ & 15) == 0; };
    return a.ldc % 8 == 0 && al16(a.C) && (!(a.epi & ST_EPI_RESIDUAL) || (a.ldr % 8 == 0 && al16(a.residual))) && al16(a.bias) && al16(a.ln_c) && al16(a.ln_d);
}

template <typename T, int BN, int WGM, int WGN>
static void gemm8p_launch(const GemmArgs& a, hipStream_t st) {
    const bool geglu = a.epi & ST_EPI_GEGLU;
    if constexpr (sizeof(T) == 2) {
        if (gemm8p_direct<T>(a, 256 / WGM / 16 >= 8)) {      // (tall = 128-row wave tiles: TM = 8)
            if (a.ln_c) { if (geglu) gemm8p_go<T, true, true, BN, WGM, WGN, true>(a, st); else gemm8p_go<T, false, true, BN, WGM, WGN, true>(a, st); }
            else { if (geglu) gemm8p_go<T, true, false, BN, WGM, WGN, true>(a, st); else gemm8p_go<T, false, false, BN, WGM, WGN, true>(a, st); }
            return;
        }
    }
    if (a.ln_c) { if (geglu) gemm8p_go<T, true, true, BN, WGM, WGN>(a, st); else gemm8p_go<T, false, true, BN, WGM, WGN>(a, st); }
    else { if (geglu) gemm8p_go<T, true, false, BN, WGM, WGN>(a, st); else gemm8p_go<T, false, false, BN, WGM, WGN>(a, st); }
}
