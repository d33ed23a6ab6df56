// =============================================================================
// gemm4w: 256 x 256 tiles on FOUR waves - one per SIMD, wave tile 128 x 128 (a developer kernel: included by csrc/dispatch.h in -DST_DEV_CONFIGS builds only).
//
// Why: the eight-phase kernel (gemm8p, wave tiles of 128 x 64) moves 192 KB of fragments out of LDS and 64 KB of DMA
// into it per K tile: 256 KB at the LDS's 128 B/clk = 2048 cycles, exactly the tile's MFMA time - it is paced by LDS
// bandwidth (matrix pipe 60 % busy in the loop).  A 128 x 128 wave tile reads every fragment once per TWO accumulator
// columns more: 128 KB per K tile.  The price is the register file: 256 accumulator registers per lane, so one wave per
// SIMD and nobody to hide its LDS latency but the wave itself - fragment reads and DMAs of the NEXT phase are interleaved
// with the MFMAs of this one (pinned with sched_group_barrier).
//
//   * LDS: two K tiles, each four regions of 128 rows x 128 B (A0, A1, B0, B1: the first / second 64 rows of both wave
//     rows, the first / second 64 columns of both wave columns), filled by LDS-DMA, 4 KiB per wave and region.
//   * A K tile is four phases over the quadrants of the wave tile, each changing ONE operand half:
//         ph0 (A0,B0)  ph1 (A0,B1)  ph2 (A1,B1)  ph3 (A1,B0)
//     and every phase reads the half the NEXT phase is first to need into a free register buffer:
//         ph0: B1(t)   ph1: A1(t)   ph2: A0(t+1)   ph3: B0(t+1)
//     A halves alternate between two buffers; the two B buffers trade roles every K tile (B0(t+1) lands where B1(t) was),
//     so the loop body is written for both parities.
//   * Two barriers per K tile (before ph0 and before ph2).  The regions read in the two phases before a barrier are
//     free after it and are refilled with the tile two ahead during the next two phases; what is read in the two phases
//     after a barrier was issued six phases (1.5 K tiles, ~3000 cycles) earlier: counted vmcnt(16), never 0.
//   * Past the last K tile the DMAs fetch one zero line (every lane the same 16 bytes) so the counts stay uniform.
// Epilogue, LayerNorm fold, GEGLU column order [values | gates], statistics, tile order: those of gemm8p.
// =============================================================================
template <typename T, bool GEGLU, bool LNF>
__global__ __launch_bounds__(256) void gemm4w_kernel(const GemmArgs p) {
    static_assert(sizeof(T) <= 2, "16-bit elements (bf16 / f16) or e4m3 bytes");
    constexpr int BM = 256, BN = 256, WGM = 2, WGN = 2, TM = 8, TN = 8, TH = 4;
    constexpr int KB = 128 / (int)sizeof(T);           // elements per 128-byte row of a K tile: 64, or 128 e4m3
    constexpr int ES = (int)sizeof(T);
    typedef typename OutT<T>::type TO;
    constexpr int HB = 128 * 128;                      // bytes of a region
    constexpr int TILE_B = 4 * HB;                     // A0 A1 B0 B1
    constexpr int BNO = GEGLU ? BN / 2 : BN;
    typedef typename Mma<T>::Frag Frag;
    typedef typename std::conditional<sizeof(T) == 1, u32x4, Frag>::type Half;
    constexpr int NKK = sizeof(T) == 1 ? 1 : 2;        // MFMA k steps per K tile
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const lnrows = lds + 2 * TILE_B;            // LayerNorm (mean, rstd) per row

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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
    const char* const zeros = reinterpret_cast<const char*>(g_zero16);

    // ---- DMA sources.  Piece e (0..3) of this wave in a region covers region rows idx = 32 * wave + 8 * e + lr:
    //      wave row / column idx / 64 (= wave >> 1), row inside its half idx % 64.  Everything but the lane's own row and
    //      16-byte chunk is wave-uniform: base pointers in SGPRs, one 32-bit lane offset per operand.
    //      The four region pointers walk along K by `kstep` bytes per use; past the last K tile they stand on the zero line.
    const int lr = lane >> 3;
    const int lc = (lane & 7) ^ lr;                  // logical 16-byte chunk this lane fetches (source-side swizzle)
    const int prow = (wave & 1) * 32 + lr;           // row inside the 64-row half, piece 0
    unsigned a_vo[4], b_vo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a_vo[e] = (unsigned)((size_t)(prow + 8 * e) * p.lda * ES + lc * 16);
        b_vo[e] = (unsigned)((size_t)(prow + 8 * e) * p.K * ES + lc * 16);
    }
    const char* rp[4];                                // A0 A1 B0 B1
    rp[0] = (const char*)p.A + ((size_t)m0 + (wave >> 1) * 128) * p.lda * ES;
    rp[1] = rp[0] + (size_t)64 * p.lda * ES;
    // tile column c -> row of W: c (plain), or value row c / gate row N + c - 128 (GEGLU: wave column 0 = values, 1 = gates)
    rp[2] = (const char*)p.W + (GEGLU ? ((wave >> 1) ? (size_t)p.N + n0 : (size_t)n0) : (size_t)n0 + (wave >> 1) * 128) * p.K * ES;
    rp[3] = rp[2] + (size_t)64 * p.K * ES;
    const int nk = p.K / KB;
    int kstep = 128;
    const unsigned dma_dst = lds_addr_of(lds) + wave * 4096;

    // region r of a K tile: 0 = A0, 1 = A1, 2 = B0, 3 = B1; four DMAs per wave (piece e), then the region's pointer moves one K tile on
    auto issue_piece = [&](int stage, int region, int e) {
        dma16_at<0>(rp[region] + (region < 2 ? a_vo[e] : b_vo[e]), dma_dst + stage * TILE_B + region * HB + e * 1024);
        if (e == 3) rp[region] += kstep;
    };
    auto issue_region = [&](int stage, int region) {
#pragma unroll
        for (int e = 0; e < 4; ++e) issue_piece(stage, region, e);
    };
    // from here on every DMA fetches the zero line (all lanes the same 16 bytes): the counts of the waits stay uniform
    auto dry_up = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a_vo[e] = 0u; b_vo[e] = 0u; }
#pragma unroll
        for (int r = 0; r < 4; ++r) rp[r] = zeros;
        kstep = 0;
    };

    // touch the epilogue's operands now (they are first read after the K loop, where a miss would be exposed)
    unsigned int touch_sink = 0;
    {
        auto touch_at = [&](const char* a) {
            a = (const char*)((uintptr_t)a & ~(uintptr_t)3);
            asm volatile("global_load_dword %0, %1, off" : "+v"(touch_sink) : "v"(a) : "memory");
        };
        auto touch = [&](const void* base, long byte_off, int nbytes) {
            for (int o = t * 128; o < nbytes; o += 256 * 128) touch_at((const char*)base + byte_off + o);
        };
        if (p.epi & ST_EPI_BIAS) {
            touch(p.bias, (long)n0 * 2, BNO * 2);
            if (GEGLU) touch(p.bias, ((long)p.N + n0) * 2, BNO * 2);
        }
        if (LNF) {
            touch(p.ln_c, (long)n0 * 4, BNO * 4); touch(p.ln_d, (long)n0 * 4, BNO * 4);
            if (GEGLU) { touch(p.ln_c, ((long)p.N + n0) * 4, BNO * 4); touch(p.ln_d, ((long)p.N + n0) * 4, BNO * 4); }
        }
        if (p.epi & ST_EPI_RESIDUAL) {
            constexpr int lines = (BNO * 2 + 127) / 128;
            for (int o = t; o < BM * lines; o += 256) {
                const int r = o / lines, l = o - r * lines;
                touch_at((const char*)p.residual + ((size_t)(m0 + r) * p.ldr + n0) * 2 + l * 128);
            }
        }
    }
    // LayerNorm-folded GEMM: row statistics from the producer's partials (one thread per row)
    LnRowSum<1> ln_sum;
    if constexpr (LNF) ln_sum.load(reinterpret_cast<const float2*>(p.ln_stats) + (size_t)(m0 + t) * p.ln_chunks, p.ln_chunks, 0);
    // prologue: both K tiles of the ring, in the order they are needed
    issue_region(0, 0); issue_region(0, 2); issue_region(0, 3); issue_region(0, 1);
    issue_region(1, 0); issue_region(1, 2); issue_region(1, 3); issue_region(1, 1);      // (K >= 2 tiles: gemm4w_applies)
    if (nk == 2) dry_up();
    if constexpr (LNF) {
        float a1, a2;
        ln_sum.finish(reinterpret_cast<const float2*>(p.ln_stats) + (size_t)(m0 + t) * p.ln_chunks, p.ln_chunks, 0, a1, a2);
        const float mean = a1 / (float)p.K;
        const float rstd = rsqrtf(fmaxf(a2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
        reinterpret_cast<float2*>(lnrows)[t] = make_float2(mean, rstd);
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, q = lane >> 4;
    // fragment addresses inside a region: row = w * 64 + frag * 16 + r16, chunk 4*kk + q, swizzled by row & 7 = r16 & 7
    // (one address register per stage: the immediate offset of ds_read reaches 64 KiB, the ring is twice that)
    const char* a_ad[2][2];
    const char* b_ad[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = ((4 * kk + q) ^ (r16 & 7)) << 4;
            a_ad[s][kk] = lds + s * TILE_B + (wm * 64 + r16) * 128 + sw;
            b_ad[s][kk] = lds + s * TILE_B + 2 * HB + (wn * 64 + r16) * 128 + sw;
        }
    Frag fa[2][TH][NKK], fb[2][TH][NKK];
    // four of the eight 16-byte reads of a phase (c = 0, 1), into half `half` (0 / 1 = first / second 64 rows (A) or columns
    // (B) of the wave tile) of buffer dst.  16-bit: k step c of all four fragments; e4m3: both halves of fragments 2c, 2c+1
    // (the 128-k operand is assembled where they land).
    auto read_four = [&](Frag (&dst)[TH][NKK], const char* const (&ad)[2], int half, int c) {
        if constexpr (sizeof(T) == 1) {
#pragma unroll
            for (int i = 2 * c; i < 2 * c + 2; ++i) {
                const int o = half * HB + i * 2048;
                const Half lo = *reinterpret_cast<const Half*>(ad[0] + o), hi = *reinterpret_cast<const Half*>(ad[1] + o);
                dst[i][0] = Frag{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            }
        } else {
#pragma unroll
            for (int i = 0; i < TH; ++i) dst[i][c] = *reinterpret_cast<const Frag*>(ad[c] + half * HB + i * 2048);
        }
    };
    // One phase = four chunks of [a quarter of the quadrant's MFMAs] [fragment reads for the next phase: all eight in the
    // first two chunks] [one DMA], fenced from each other and pinned in that order.  hipcc waits for lgkmcnt(0) in front of
    // the first MFMA that uses a new buffer, whatever has been requested since: with the reads BEHIND the MFMAs of their chunk
    // and none in the last two chunks that wait finds nothing outstanding.  The four or five other instructions of a chunk
    // fit the issue slots its last MFMA leaves free while it occupies the pipe.
    auto phase = [&](const Frag (&a)[TH][NKK], const Frag (&b)[TH][NKK], auto mh_, auto nh_,
                     Frag (&rd)[TH][NKK], const char* const (&ad)[2], int half, int stage, int region) {
        constexpr int mh = decltype(mh_)::value, nh = decltype(nh_)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if constexpr (NKK == 2) {
                const int kk = c >> 1;
#pragma unroll
                for (int i = 2 * (c & 1); i < 2 * (c & 1) + 2; ++i)
#pragma unroll
                    for (int j = 0; j < TH; ++j) Mma<T>::run(acc[mh * TH + i][nh * TH + j], b[j][kk], a[i][kk]);
            } else {
#pragma unroll
                for (int j = 0; j < TH; ++j) Mma<T>::run(acc[mh * TH + c][nh * TH + j], b[j][0], a[c][0]);
            }
            if (c < 2) read_four(rd, ad, half, c);
            issue_piece(stage, region, c);
            __builtin_amdgcn_sched_group_barrier(0x008, NKK == 2 ? 8 : 4, 0);
            if (c < 2) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // lgkmcnt(0) vmcnt(16) through the builtin, so that hipcc's own wait bookkeeping knows the fragments have arrived
#define ST4W_SYNC()                                       \
    __builtin_amdgcn_sched_barrier(0);                    \
    __builtin_amdgcn_s_waitcnt(0x4070);                   \
    __builtin_amdgcn_s_barrier();                         \
    __builtin_amdgcn_sched_barrier(0)

    wait_vmcnt<24>();                                // A0(0), B0(0) have landed (and every load older than the DMAs)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::"v"(touch_sink));
#pragma unroll
    for (int c = 0; c < 2; ++c) { read_four(fa[0], a_ad[0], 0, c); read_four(fb[0], b_ad[0], 0, c); }

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    auto ktile = [&](int kt, auto par_) {
        constexpr int P = decltype(par_)::value;
        ST4W_SYNC();                                 // B1(t), A1(t) landed; A0 / B0 regions of this tile are free
        phase(fa[0], fb[P], I0{}, I0{}, fb[P ^ 1], b_ad[P], 1, P, 0);              // reads B1(t), requests A0(t+2)
        phase(fa[0], fb[P ^ 1], I0{}, I1{}, fa[1], a_ad[P], 1, P, 2);              // reads A1(t), requests B0(t+2)
        ST4W_SYNC();                                 // A0(t+1), B0(t+1) landed; B1 / A1 regions of this tile are free
        phase(fa[1], fb[P ^ 1], I1{}, I1{}, fa[0], a_ad[P ^ 1], 0, P, 3);          // reads A0(t+1), requests B1(t+2)
        phase(fa[1], fb[P], I1{}, I0{}, fb[P ^ 1], b_ad[P ^ 1], 0, P, 1);          // reads B0(t+1), requests A1(t+2)
        if (kt + 3 == nk) dry_up();                  // the next tile would request K tile nk
    };
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(kt, I0{});
        if (kt + 1 < nk) ktile(kt + 1, I1{});
    }
#undef ST4W_SYNC
    wait_vmcnt<0>();                                  // no LDS-DMA may outlive the workgroup's LDS allocation
    staged_epilogue<TO, BM, BN, WGM, WGN, TM, TN, GEGLU, 2 * TILE_B, !LNF, sizeof(T) == 1>(p, acc, m0, n0, tile_n, wm, r16, q, ColsPlain{wn, 128}, lds,
                                                                        reinterpret_cast<const float2*>(lnrows));
}

static inline bool gemm4w_applies(const GemmArgs& a, int kb = 64) {
    const long n_rows = (a.epi & ST_EPI_GEGLU) ? 2L * a.N : a.N;
    return a.M % 256 == 0 && n_rows % 256 == 0 && a.K % kb == 0 && a.K >= 2 * kb && a.N % 8 == 0 &&
           !(a.epi & ST_EPI_ROWBIAS) && (!a.row_stats || !(a.epi & ST_EPI_GEGLU)) &&
           (size_t)256 * a.lda * 2 < (1ull << 31) && (size_t)256 * a.K * 2 < (1ull << 31);
}

template <typename T, bool GEGLU, bool LNF>
static void gemm4w_go(const GemmArgs& a, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)(4 * 128 * 128) + 256 * 8;
    auto kfn = gemm4w_kernel<T, GEGLU, LNF>;
    static unsigned long long lds_ok = 0;
    ensure_dynamic_lds(kfn, lds, &lds_ok);
    GemmArgs b = a;
    const int tiles_m = a.M / 256, tiles_n = (int)(((a.epi & ST_EPI_GEGLU) ? 2L * a.N : a.N) / 256);
    {   // XCD partition of the tile order: bytes from beyond L2 ~ A * (8 / panels) + W * panels
        const double abytes = (double)a.M * a.K, wbytes = (double)tiles_n * 256 * a.K;
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
    hipLaunchKernelGGL(kfn, dim3(main_blocks + b.helper_blocks), dim3(256), lds, st, b);
}

template <typename T>
static void gemm4w_launch(const GemmArgs& a, hipStream_t st) {
    const bool geglu = a.epi & ST_EPI_GEGLU;
    if (a.ln_c) { if (geglu) gemm4w_go<T, true, true>(a, st); else gemm4w_go<T, false, true>(a, st); }
    else { if (geglu) gemm4w_go<T, true, false>(a, st); else gemm4w_go<T, false, false>(a, st); }
}
