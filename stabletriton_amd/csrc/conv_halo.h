// GEMM-shaped operators: conv3x3 / stride 1 with the input patch resident in LDS (16-bit elements; split fp32 operands of the
// strict mode: channel slices of 32, three MFMAs per product, two accumulator sets) and its host side.
// Internal to csrc/.
#pragma once
#include "gemm_dma.h"

// =============================================================================
// conv3x3, stride 1, pad 1, with the input patch resident in LDS ("halo" loop).
// The K loop runs channel-slice-major: for every 64 input channels the (TH+2) x (W+2)
// pixel patch that the block's TH full image rows need is staged ONCE (zero padding
// included) and all nine taps read their A fragments from it at a pixel offset;
// only the weights stream per tap (3-deep ring).  The implicit-GEMM loop above
// fetches the same input pixels once per tap: for the 256x128 tiles this one issues
// 2.3x fewer DMA pieces, which is what bounds these kernels.
// Tile: BM = TH * W = 256 output pixels (TH full rows of one image) x BN = 128 channels,
// 8 waves (4 x 2, 64 x 64 wave tiles); epilogue and in-launch split-K (over channel
// slices) are shared with gemm_dma_kernel.
// =============================================================================
// UPS: the nearest-2x upsample folded in (W, TH count OUTPUT pixels; the patch holds INPUT pixels: output
// (oy, ox) tap (r, s) reads input ((oy + r - 1) >> 1, (ox + s - 1) >> 1), zero outside).
template <typename T, int WL2, int TH, int BN, int WGM, int WGN, bool UPS = false>
__global__ __launch_bounds__(512) void conv_halo_kernel(const GemmArgs p) {
    static_assert(sizeof(T) == 2 || is_split<T>(), "16-bit elements (bf16 / f16) or split fp32 operands");
    constexpr int W = 1 << WL2, BM = TH * W, NW = WGM * WGN;
    constexpr int VEC = 16 / (int)sizeof(T);                         // elements per 16-byte chunk
    typedef typename OutT<T>::type TO;                               // element type of y, bias, residual (split operands: fp32)
    static_assert(NW == 8 && BM % (16 * WGM) == 0 && BN % (16 * WGN) == 0 && W >= 16, "tile / wave layout");
    static_assert(!UPS || TH % 2 == 0, "upsampled tiles start on an even output row");
    constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 16, TN = WTN / 16, KB = 128 / (int)sizeof(T);      // a channel slice = 128 bytes per pixel
    constexpr int WI = UPS ? W / 2 : W;                              // input image width
    constexpr int PROWS = UPS ? TH / 2 + 2 : TH + 2;                 // input rows the tile touches
    constexpr int PWD = WI + 2, PPX = PROWS * PWD;                   // patch row pitch and pixel count
    constexpr int PIECES_P = (PPX + 7) / 8, PB = PIECES_P * 1024;    // 1-KiB pieces (8 pixels x 128 B) of one patch
    constexpr int PWV = (PIECES_P + NW - 1) / NW;                    // patch pieces per wave per channel slice
    constexpr int STAGES = 3, WT_B = BN * 128;                       // weight ring
    constexpr int B_PIECES = BN / 8, B_IT = (B_PIECES + NW - 1) / NW;   // weight pieces (per wave per trip)
    typedef typename Mma<T>::Frag Frag;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const ring = lds + 2 * PB;
    char* const dump = ring + STAGES * WT_B;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int r16 = lane & 15, q = lane >> 4;
    const int lr = lane >> 3;
    const int Hh = p.Hin;                            // input image height
    const int tiles_m = p.M / BM;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    // (divisions by launch constants as multiply-high, see GemmArgs::tm_*: here tm_mg_per_panel is the magic of tiles_m and
    //  tm_mg_rows that of the tiles per image)
    const int tw = mg_div(wg, p.tm_mg_splitk), split = wg - tw * p.splitk;
    const int tile_n = mg_div(tw, p.tm_mg_per_panel), tile_m = tw - tile_n * tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int rows_per_img = p.Hout / TH;
    const int img = mg_div(tile_m, p.tm_mg_rows), ty0 = (tile_m - img * rows_per_img) * TH;      // first OUTPUT row of the tile
    const int iy_base = UPS ? (ty0 >> 1) - 1 : ty0 - 1;                                    // input row of patch row 0
    const int cs_lo = split * p.nk_base + min(split, p.nk_rem), cs_hi = cs_lo + p.nk_base + (split < p.nk_rem ? 1 : 0);

    const T* __restrict__ Xb = (const T*)p.A + (size_t)img * Hh * WI * p.Cin;
    const T* __restrict__ Wp = (const T*)p.W;
    const T* zeros = reinterpret_cast<const T*>(g_zero16);

    // ---- per-lane DMA sources -------------------------------------------------------------------
    const T* pa_ptr[PWV];                           // patch pixel of piece e (channel slice 0), or null = zero fill
#pragma unroll
    for (int e = 0; e < PWV; ++e) {
        const int pidx = e * NW + wave;
        const int pp = pidx * 8 + lr;
        const int py = pp / PWD, px = pp - py * PWD;
        const int y = iy_base + py, x = px - 1;
        const bool ok = pp < PPX && y >= 0 && y < Hh && x >= 0 && x < WI;
        const int lc = (lane & 7) ^ (pp & 7);
        pa_ptr[e] = ok ? Xb + ((size_t)y * WI + x) * p.Cin + lc * VEC : nullptr;
    }
    const T* pb_ptr[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int row = (wave + j * NW) * 8 + lr;
        const int wrow = n0 + row;
        pb_ptr[j] = (wrow < p.N && wave + j * NW < B_PIECES) ? Wp + (size_t)wrow * p.K + ((lane & 7) ^ lr) * VEC : nullptr;
    }
    auto issue_patch = [&](int cs, int e) {         // piece e of this wave, channel slice cs (cs >= cs_hi: dummy)
        const int pidx = e * NW + wave;
        const bool live = pidx < PIECES_P && cs < cs_hi;
        const T* src = (live && pa_ptr[e]) ? pa_ptr[e] + cs * KB : zeros;
        dma16_at<0>(src, live ? lds_addr_of(lds) + (cs & 1) * PB + pidx * 1024 : lds_addr_of(lds) + 2 * PB + STAGES * WT_B);      // (= dump)
    };
    auto issue_w = [&](int cs, int tap, int slot) {  // the weight tile of trip (cs, tap); cs >= cs_hi: dummy
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            // (BN = 160: twenty pieces over eight waves - the waves without a third piece issue a dummy, which keeps
            // the trip straight-line code with one vmcnt for all waves; measured faster than a per-wave branch)
            const bool live = cs < cs_hi && wave + j * NW < B_PIECES;
            const T* src = (live && pb_ptr[j]) ? pb_ptr[j] + (size_t)tap * p.Cin + cs * KB : zeros;
            dma16_at<0>(src, live ? lds_addr_of(lds) + 2 * PB + slot * WT_B + (wave + j * NW) * 1024 : lds_addr_of(lds) + 2 * PB + STAGES * WT_B);
        }
    };

    // ---- fragment addresses -----------------------------------------------------------------------
    int pp0[TM];                                     // plain: patch pixel of tap (0,0); UPS: (output row, column) packed
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 16 + r16;
        pp0[i] = UPS ? (((row >> WL2) << 16) | (row & (W - 1))) : (row >> WL2) * PWD + (row & (W - 1));
    }
    int rowb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) rowb[j] = wn * WTN + j * 16 + r16;

    f32x4 acc[TM][TN];
    f32x4 corr[is_split<T>() ? TM : 1][is_split<T>() ? TN : 1];      // split operands: the cross products, in units of 2^-11
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (is_split<T>()) corr[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    // ---- prologue: the first patch, the first two weight tiles ------------------------------------------
#pragma unroll
    for (int e = 0; e < PWV; ++e) issue_patch(cs_lo, e);
    issue_w(cs_lo, 0, 0);
    issue_w(cs_lo, 1, 1);
    wait_vmcnt<B_IT>();                              // all but the second weight tile have landed
    __builtin_amdgcn_s_barrier();

    int slot = 0;                                    // ring slot of the current trip
    for (int cs = cs_lo; cs < cs_hi; ++cs) {
        const char* patch = lds + (cs & 1) * PB;
        auto trip = [&](auto tc) {
            constexpr int tap = decltype(tc)::value;
            constexpr int r = tap / 3, s_ = tap - r * 3;
            // patch pieces of the NEXT channel slice go out with taps 0..4 (its buffer was last read in the
            // previous slice), then the weight tile two trips ahead (its slot was read in the previous trip)
            constexpr int n_p = tap < 5 ? PWV / 5 + (tap < PWV % 5 ? 1 : 0) : 0;
            constexpr int p_lo = tap < 5 ? tap * (PWV / 5) + (tap < PWV % 5 ? tap : PWV % 5) : PWV;
#pragma unroll
            for (int e = 0; e < n_p; ++e) issue_patch(cs + 1, p_lo + e);
            {
                constexpr int tap2 = (tap + 2) % 9;
                const int slot2 = slot >= 1 ? slot - 1 : 2;          // (slot + 2) % 3
                issue_w(tap + 2 >= 9 ? cs + 1 : cs, tap2, slot2);
            }
            const char* wt = ring + slot * WT_B;
            // (the row indices pass through an empty asm every trip: otherwise hipcc hoists all 9 x 2 x 8
            // fragment addresses out of the channel-slice loop and spills)
            int ppl[TM], rbl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) { ppl[i] = pp0[i]; asm volatile("" : "+v"(ppl[i])); }
#pragma unroll
            for (int j = 0; j < TN; ++j) { rbl[j] = rowb[j]; asm volatile("" : "+v"(rbl[j])); }
            // every fragment of the trip is read up front (16-bit: both 32-wide K halves, the second half's LDS latency hides
            // under the first half's MFMAs; split operands: the hi chunk q and the lo chunk q + 4 of the one 32-wide step)
            constexpr int NG_ = frag2<T>() ? 1 : 2;
            Frag fa[NG_][TM], fb[NG_][TN];
            auto rd = [&](const char* base, int row, int g) -> Frag {
                if constexpr (frag2<T>()) {
                    const u32x4 lo = *reinterpret_cast<const u32x4*>(base + row * 128 + ((q ^ (row & 7)) << 4));
                    const u32x4 hi = *reinterpret_cast<const u32x4*>(base + row * 128 + (((q + 4) ^ (row & 7)) << 4));
                    return Frag{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                } else {
                    const int c = 4 * g + q;
                    return *reinterpret_cast<const Frag*>(base + row * 128 + ((c ^ (row & 7)) << 4));
                }
            };
#pragma unroll
            for (int g = 0; g < NG_; ++g) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    int pp;
                    if constexpr (UPS) {
                        const int uy = ty0 + (ppl[i] >> 16) + r - 1, ux = (ppl[i] & 0xffff) + s_ - 1;
                        pp = ((uy >> 1) - iy_base) * PWD + (ux >> 1) + 1;
                    } else {
                        pp = ppl[i] + r * PWD + s_;
                    }
                    fa[g][i] = rd(patch, pp, g);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[g][j] = rd(wt, rbl[j], g);
            }
#pragma unroll
            for (int g = 0; g < NG_; ++g)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (is_split<T>()) Mma<T>::run2(acc[i][j], corr[i][j], fb[g][j], fa[g][i]);
                        else Mma<T>::run(acc[i][j], fb[g][j], fa[g][i]);
                    }
            // pin the emitted order: DMAs, then every fragment read, then the MFMAs -
            // left alone hipcc sinks each read to just before its first use and waits lgkmcnt(0) a dozen times per trip
            __builtin_amdgcn_sched_group_barrier(0x020, n_p + B_IT, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NG_ * TM * TN * mfma_per_frag<T>(), 0);
            __builtin_amdgcn_sched_barrier(0);
            // the next trip's weight tile (issued one trip ago, before this trip's DMAs) must have landed - and
            // with it, in issue order, every patch piece of the next slice
            wait_vmcnt<n_p + B_IT>();
            __builtin_amdgcn_s_barrier();
            slot = slot == 2 ? 0 : slot + 1;
        };
        trip(std::integral_constant<int, 0>{}); trip(std::integral_constant<int, 1>{}); trip(std::integral_constant<int, 2>{});
        trip(std::integral_constant<int, 3>{}); trip(std::integral_constant<int, 4>{}); trip(std::integral_constant<int, 5>{});
        trip(std::integral_constant<int, 6>{}); trip(std::integral_constant<int, 7>{}); trip(std::integral_constant<int, 8>{});
    }
    wait_vmcnt<0>();                                 // no LDS-DMA may outlive the workgroup's LDS allocation
    __builtin_amdgcn_s_barrier();
    if constexpr (is_split<T>()) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = __builtin_fmaf(corr[i][j][e], ST_SPLIT_INV, acc[i][j][e]);
    }
    if (p.splitk > 1) {
        if (!splitk_combine<TM, TN, BM * BN>(p, acc, tw, split, lds, t, wave, lane)) {
            unsigned int sink = 0;                   // this block is done: its slice of the next weights, then exit
            touch_next_weights(p, sink);
            retire_touches(sink);
            return;
        }
    }
    staged_epilogue<TO, BM, BN, WGM, WGN, TM, TN, false, 2 * PB + STAGES * WT_B, true>(p, acc, m0, n0, tile_n, wm, r16, q, ColsPlain{wn, WTN}, lds,
                                                                                      nullptr);
}

// ---- host side of conv_halo_kernel --------------------------------------------------------------
static inline bool conv_halo_applies(const GemmArgs& a, int R, int ups, int kb = 64) {      // kb: channels per slice (64 16-bit, 32 split fp32)
    static const bool off = dev_env_int("ST_CONV_HALO", 1) == 0;
    if (off) return false;
    if (R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1) return false;
    if (a.Cin % kb != 0 || a.N % 4 != 0 || a.N < 64) return false;
    if (ups) {           // output 64 or 128 pixels wide, tiles of 256 output pixels
        if (a.Wout != 2 * a.Win || a.Hout != 2 * a.Hin || (a.Wout != 64 && a.Wout != 128)) return false;
        return a.Hout % (256 / a.Wout) == 0 && a.M % 256 == 0;
    }
    // (16 x 16: the SDXL-refiner's fourth level, one whole image per tile; 16-bit elements only)
    if (a.Win != 32 && a.Win != 64 && a.Win != 128 && !(a.Win == 16 && kb == 64)) return false;
    const int bm = a.Win == 128 ? 128 : 256, th = bm / a.Win;
    return a.Hin == a.Hout && a.Win == a.Wout && a.Hin % th == 0 && a.M % bm == 0;
}

template <typename T, int WL2, int TH, int BN, int WGM, int WGN, bool UPS = false>
static void conv_halo_go(const GemmArgs& b, int blocks, hipStream_t st) {
    constexpr int W = 1 << WL2;
    constexpr int PPX = UPS ? (TH / 2 + 2) * (W / 2 + 2) : (TH + 2) * (W + 2);
    constexpr size_t lds = 2 * (size_t)((PPX + 7) / 8) * 1024 + 3 * BN * 128 + 1024;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kfn = conv_halo_kernel<T, WL2, TH, BN, WGM, WGN, UPS>;
    static unsigned long long lds_ok = 0;
    ensure_dynamic_lds(kfn, lds, &lds_ok);
    GemmArgs c = b;
    if (!colstats_ok(b, TH * W, false)) c.col_stats = nullptr;
    c.helper_blocks = 0;
    fill_next_per(c, blocks);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, st, c);
}

template <typename T>
static int conv_halo_launch(const GemmArgs& a, hipStream_t st) {
    GemmArgs b = a;
    // 256 pixels x 128 channels at the 32- and 64-pixel levels; one 128-pixel row x 160 channels at the 128-pixel level
    const bool ups = a.ups != 0;
    // split operands carry two accumulator sets (256 registers at 128 channels per tile): 128-channel tiles where the output is
    // at most 64 pixels wide and the channels divide (640 @ 64 x 64: 124 -> 89 us), 64-channel tiles elsewhere (320 @ 128 x 128:
    // 109 against 134 us)
    constexpr bool SP = is_split<T>();
    const bool sp128 = SP && a.N % 128 == 0 && a.Wout <= 64;
    const int bm = (!ups && a.Win == 128) ? 128 : 256;
    int bn = SP ? (sp128 ? 128 : 64) : ((ups || a.Win == 128) ? 160 : 128);
    if (!SP && !ups && a.Win == 128) {
        // 128-pixel rows: 160 channels per tile (320 = 2 x 160: SDXL-base's first level is one round of 256 tiles at batch 1), or
        // 128 where that takes fewer rounds x tile width (the refiner's 384 channels: 3 x 128 instead of 160 + 160 + a ragged 64)
        auto cost = [&](int w) { return (long)cdiv((long)(a.M / bm) * cdiv(a.N, w), 256L) * w; };
        if (cost(128) < cost(160)) bn = 128;
    }
    const int tiles = (a.M / bm) * cdiv(a.N, bn);
    const int ncs = a.Cin / (128 / (int)sizeof(T));
    // K split over channel slices (a slice keeps at least two channel slices), by a time model fitted to the sweeps of
    // tools/conv_splitk_sweep.py (1280 @ 32 x 32 at batch 1: 144 / 83 / 65 / 55 / 50 / 51 us for 1 ... 6 slices):
    //   rounds of 256 blocks x (tap trips per slice x 0.77 us + 5.4 us of prologue and epilogue) + per slice 1.2 us and its fp32
    //   slab written and read once at ~4 TB/s.
    // At batch 1 this is the rule of rounds 2-3 (one round of ~240 blocks: 40 tiles -> 6 slices, 80 -> 3, 128 -> 2); at batch 2 / 4
    // it no longer picks the splits that only add a second, mostly empty round (160 tiles x 2 slices = 320 blocks: the step at
    // batch 4 35.66 -> 34.89 ms, at batch 2 20.88 -> 20.61, same box, tools/ab_step.py; per shape profiles/r04_conv_splits.txt).
    // Split operands (strict mode): the same with its own two constants; at batch 1 the old choices, at batch 4 three slices where
    // the old rule took two (1280 @ 32 x 32: 432 -> 343 us).
    int sk = 1;
    if (a.partial && tiles <= 16384) {
        static const int force = dev_env_int("ST_HALO_SPLITS", 0);          // dev knob: this many slices where allowed
        // (split operands: 32-channel slices, three MFMAs per product - the trip costs about the same, the fixed part about twice)
        const double trip_us = (SP ? 0.75 : 0.77) * bn / 128.0, fixed_us = SP ? 10.0 : 5.4, slab_us = 2.0 * tiles * bm * bn * 4.0 / 4.0e6;
        const int trips = ncs * 9, cap = tiles <= 16 ? 16 : 8, max_sk = ncs / 2 < cap ? ncs / 2 : cap;      // (a dozen tiles: the 16 x 16 level)
        double best = 1e30;
        for (int s_ = 1; s_ <= max_sk || s_ == 1; ++s_) {
            if (s_ > 1 && (size_t)s_ * tiles * bm * bn * 4 + 65536 > a.partial_bytes) break;
            const int rounds = (tiles * s_ + 255) / 256;
            const double t = rounds * (trips / (double)s_ * trip_us + fixed_us) + (s_ > 1 ? s_ * (1.2 + slab_us) : 0.0);
            if (force ? s_ == force : t < best - 0.5) { best = t; sk = s_; }      // (near ties: the fewer slices)
        }
    }
    if (sk > 1) { b.splitk = sk; b.tile_counters = (int*)a.partial; b.partial = a.partial + 16384; }
    else b.splitk = 1;
    {   // the kernel's divisions as multiply-high (GemmArgs::tm_*): by the K slices, the tile rows, the tiles per image
        const int tiles_m = a.M / bm, th = bm / a.Wout;
        b.tm_mg_splitk = magic_u32((unsigned)b.splitk);
        b.tm_mg_per_panel = magic_u32((unsigned)tiles_m);
        b.tm_mg_rows = magic_u32((unsigned)(a.Hout / th));
        b.nk_base = ncs / b.splitk; b.nk_rem = ncs % b.splitk;
    }
    b.stats_chunks = cdiv(a.N, bn);
    if (a.row_stats && b.stats_chunks > a.stats_capacity) return st_fail("conv2d: row_stats buffer holds %d chunks, %d needed", a.stats_capacity, b.stats_chunks);
    if (a.stats_chunks_out) *a.stats_chunks_out = a.row_stats ? b.stats_chunks : 0;
    if constexpr (SP) {
        if (ups && a.Wout == 64) { if (sp128) conv_halo_go<T, 6, 4, 128, 4, 2, true>(b, tiles * sk, st); else conv_halo_go<T, 6, 4, 64, 4, 2, true>(b, tiles * sk, st); }
        else if (ups) conv_halo_go<T, 7, 2, 64, 4, 2, true>(b, tiles * sk, st);
        else if (a.Win == 32) { if (sp128) conv_halo_go<T, 5, 8, 128, 4, 2>(b, tiles * sk, st); else conv_halo_go<T, 5, 8, 64, 4, 2>(b, tiles * sk, st); }
        else if (a.Win == 64) { if (sp128) conv_halo_go<T, 6, 4, 128, 4, 2>(b, tiles * sk, st); else conv_halo_go<T, 6, 4, 64, 4, 2>(b, tiles * sk, st); }
        else conv_halo_go<T, 7, 1, 64, 4, 2>(b, tiles * sk, st);
    } else {
        if (ups && a.Wout == 64) conv_halo_go<T, 6, 4, 160, 4, 2, true>(b, tiles * sk, st);
        else if (ups) conv_halo_go<T, 7, 2, 160, 4, 2, true>(b, tiles * sk, st);
        else if (a.Win == 16) conv_halo_go<T, 4, 16, 128, 4, 2>(b, tiles * sk, st);
        else if (a.Win == 32) conv_halo_go<T, 5, 8, 128, 4, 2>(b, tiles * sk, st);
        else if (a.Win == 64) conv_halo_go<T, 6, 4, 128, 4, 2>(b, tiles * sk, st);
        else if (bn == 128) conv_halo_go<T, 7, 1, 128, 4, 2>(b, tiles * sk, st);
        else conv_halo_go<T, 7, 1, 160, 4, 2>(b, tiles * sk, st);
    }
    return st_check_launch("conv2d(halo)");
}
