// GEMM-shaped operators: launchers of the LDS-DMA kernel, the tile configurations and the cost model that picks one
// (gemm_dispatch), argument checks shared by the entry points.  Internal to csrc/.
#pragma once
#include "gemm_reg.h"
#include "gemm_dma.h"
#include "gemm8p.h"
#ifdef ST_DEV_CONFIGS      // the four-wave 256 x 256 variant: developer builds only (tools/dev_kernels/, tools/build_one_variant.sh)
#include "../../tools/dev_kernels/gemm4w.h"
void gemm4w_bf16(const GemmArgs& a, hipStream_t st);      // (instantiated in tools/dev_kernels/gemm_4w.hip)
void gemm4w_f16(const GemmArgs& a, hipStream_t st);
void gemm4w_fp8(const GemmArgs& a, hipStream_t st);
template <typename T> static inline void gemm4w_call(const GemmArgs& a, hipStream_t st) {
    if constexpr (std::is_same<T, bf16>::value) gemm4w_bf16(a, st);
    else if constexpr (std::is_same<T, f16>::value) gemm4w_f16(a, st);
    else gemm4w_fp8(a, st);
}
#endif

// Can a launch with BM-row tiles emit GroupNorm partials?  (tile rows must not straddle images; the LayerNorm-folded
// kernels have no scratch for it.)  Tells the host through *col_rows_out.
static inline bool colstats_ok(const GemmArgs& a, int bm, bool lnf) {
    const bool ok = a.col_stats && !lnf && (a.N & 3) == 0 && a.rows_per_batch > 0 && a.rows_per_batch % bm == 0 &&
                    cdiv(a.M, bm) <= a.col_tiles_cap;
    if (a.col_rows_out) *a.col_rows_out = ok ? bm : 0;
    return ok;
}

template <typename T, int BM, int BN, int WGM, int WGN, int STAGES, int U, bool CONV, bool GEGLU, bool LNF, bool XA = false>
static void launch_dma_one(const GemmArgs& a, hipStream_t st, int tiles_n) {
    const size_t lds = (size_t)STAGES * U * (BM + BN) * 128 + (size_t)BM * 8 + 1024;      // ring + LayerNorm (mean, rstd) per row + DMA dump
    const int sk = a.splitk > 1 ? a.splitk : 1;
    auto kfn = gemm_dma_kernel<T, BM, BN, WGM, WGN, STAGES, U, CONV, GEGLU, LNF, XA>;
    const bool emit_cols = colstats_ok(a, BM, LNF);
    static unsigned long long lds_ok = 0;
    ensure_dynamic_lds(kfn, lds, &lds_ok);
    // XCD partition: bytes from beyond L2 ~ A * (8 / panels) + W * panels (A = activations, all of K)
    GemmArgs b = a;
    {
        static const int force_pm = dev_env_int("ST_GEMM_PANELS", 0);
        const int tiles_m = cdiv(a.M, BM);
        const double abytes = CONV ? (double)a.M * a.Cin * (a.ups ? 0.25 : 1.0) * a.stride * a.stride : (double)a.M * a.K;
        const double wbytes = (double)(GEGLU ? 2 : 1) * a.N * a.K;
        int best_p = 1;
        double best = 1e300;
        for (int pm = 1; pm <= 8; pm *= 2) {
            if (pm > tiles_m) break;
            const double c = abytes * (8.0 / pm) + wbytes * pm;
            if (c < best) { best = c; best_p = pm; }
        }
        if (force_pm > 0) best_p = force_pm > tiles_m ? tiles_m : force_pm;
        b.panel_h = cdiv(tiles_m, best_p);
    }
    const int main_blocks = cdiv(a.M, BM) * tiles_n * sk;
    b.splitk = sk;
    fill_tile_map(b, cdiv(a.M, BM), tiles_n, a.K / ((128 / (int)sizeof(T)) * U));
    // launches that leave CUs idle hand the next-weights touches to helper blocks on those CUs (they run beside the K
    // loops instead of extending the epilogues)
    static const bool no_helpers = dev_env_int("ST_NO_HELPER_BLOCKS", 0) != 0;
    b.helper_blocks = (b.next_w && main_blocks <= 208 && !no_helpers) ? (256 - main_blocks > 96 ? 96 : 256 - main_blocks) : 0;
    if (!emit_cols) b.col_stats = nullptr;
    fill_next_per(b, main_blocks + b.helper_blocks);
    hipLaunchKernelGGL(kfn, dim3(main_blocks + b.helper_blocks), dim3(WGM * WGN * 64), lds, st, b);
}

template <typename T, int BM, int BN, int WGM, int WGN, int STAGES, int U, bool CONV>
static void launch_dma(const GemmArgs& a, hipStream_t st) {
    if constexpr (!CONV) {
        const bool geglu = a.epi & ST_EPI_GEGLU;
        constexpr bool PAIRS = (BN % 32 == 0);                  // GEGLU: value and gate halves of the tile are whole accumulator tiles
        if (a.ln_c) {          // LayerNorm-folded variants (never split over K: the row statistics need all of K)
            if constexpr (PAIRS) {
                if (geglu) { launch_dma_one<T, BM, BN, WGM, WGN, STAGES, U, false, true, true>(a, st, cdiv(a.N, BN / 2)); return; }
            }
            launch_dma_one<T, BM, BN, WGM, WGN, STAGES, U, false, false, true>(a, st, cdiv(a.N, BN));
            return;
        }
        if constexpr (PAIRS) {
            if (geglu) { launch_dma_one<T, BM, BN, WGM, WGN, STAGES, U, false, true, false>(a, st, cdiv(a.N, BN / 2)); return; }
        }
    }
    launch_dma_one<T, BM, BN, WGM, WGN, STAGES, U, CONV, false, false>(a, st, cdiv(a.N, BN));
}

// Tile configurations of the LDS-DMA kernel.  ST_GEMM_FORCE=<id> (developer knob)
// overrides the heuristic for A/B runs.
enum { CFG_64x64_S4 = 0, CFG_64x64_S8 = 1, CFG_64x64_S4_U2 = 2, CFG_128x64_S4 = 3, CFG_128x64_S3_U2 = 4,
       CFG_128x128_S3 = 5, CFG_64x64_S3 = 6, CFG_64x64_W8 = 7, CFG_128x64_W8 = 8, CFG_128x128_W8 = 9,
       CFG_64x128_W8 = 10, CFG_64x128_W8_S6 = 11, CFG_128x64_W8_S6 = 12, CFG_128x128_W8_S4 = 13, CFG_64x64_W8_S8 = 14, CFG_64x128_W8_U2 = 15, CFG_128x64_W8_U2 = 16, CFG_64x64_W8_U2 = 17, CFG_256x256_W8 = 18, CFG_256x128_W8 = 19, CFG_128x128_W8_S2 = 20, CFG_128x64_W8_S3 = 21, CFG_64x128_W8_S3 = 22, CFG_128x320_W8 = 23, CFG_128x256_W8 = 24, CFG_64x320_W8 = 25, CFG_64x80_W4 = 26, CFG_128x80_W8 = 27, CFG_128x160_W8 = 28, CFG_128x128_N4_S2 = 29, CFG_128x64_N4_S3 = 30, CFG_64x128_N4_S3 = 31, CFG_256x160_W8_S2 = 32, CFG_COUNT, CFG_256x256_8P = 100, CFG_256x160_8P = 101, CFG_256x256_4W = 102 };

static inline int cfg_bn(int cfg) {
    switch (cfg) {
        case CFG_64x64_S4: case CFG_64x64_S8: case CFG_64x64_S4_U2: case CFG_128x64_S4: case CFG_128x64_S3_U2: case CFG_64x64_S3:
        case CFG_64x64_W8: case CFG_128x64_W8: case CFG_128x64_W8_S6: case CFG_64x64_W8_S8: case CFG_128x64_W8_U2:
        case CFG_64x64_W8_U2: case CFG_128x64_W8_S3: case CFG_128x64_N4_S3: return 64;
        case CFG_256x256_W8: case CFG_128x256_W8: return 256;
        case CFG_128x320_W8: case CFG_64x320_W8: return 320;
        case CFG_64x80_W4: case CFG_128x80_W8: return 80;
        case CFG_128x160_W8: case CFG_256x160_W8_S2: return 160;
        default: return 128;
    }
}

#ifdef ST_DEV_CONFIGS
extern int g_dbg_cfg, g_dbg_fusek;             // (gemm_api.hip: st_debug_force_gemm)
#endif
static inline int forced_cfg() {
    static int v = dev_env_int("ST_GEMM_FORCE", -1);
#ifdef ST_DEV_CONFIGS
    if (g_dbg_cfg >= 0) return g_dbg_cfg;
#endif
    return v;
}

// (developer A/B of the model's tile choices IN the step - tools/ab_step.py against a variant built with e.g. -DST_TW_128x64=1.2:
//  the per-trip constant of that tile scaled for the 16-bit dense launches; all 1.0 in the product)
#ifndef ST_TW_128x128
#define ST_TW_128x128 1.0
#endif
#ifndef ST_TW_64x128
#define ST_TW_64x128 1.0
#endif
#ifndef ST_TW_128x64
#define ST_TW_128x64 1.0
#endif
#ifndef ST_TW_64x64
#define ST_TW_64x64 1.0
#endif
#ifndef ST_TW_128x80
#define ST_TW_128x80 1.0
#endif
#ifndef ST_TW_128x160
#define ST_TW_128x160 1.0
#endif
#ifndef ST_TW_SPLITK
#define ST_TW_SPLITK 1.0
#endif
#ifndef ST_TW_8P
#define ST_TW_8P 1.1                 // the eight-phase kernel takes the launches it prices below this x the small tiles' best
#endif
static inline double model_tweak(int cfg) {
    switch (cfg) {
        case 9: return ST_TW_128x128;      // CFG_128x128_W8
        case 10: return ST_TW_64x128;      // CFG_64x128_W8
        case 8: return ST_TW_128x64;       // CFG_128x64_W8
        case 7: return ST_TW_64x64;        // CFG_64x64_W8
        case 27: return ST_TW_128x80;      // CFG_128x80_W8
        case 28: return ST_TW_128x160;     // CFG_128x160_W8
        default: return 1.0;
    }
}
#ifndef ST_SPLIT_DENSE_TRIP
#define ST_SPLIT_DENSE_TRIP 1.0      // (developer A/B of the strict mode's tile choices: -DST_SPLIT_DENSE_TRIP=<factor> -DST_SPLIT_160_TRIP=<us>; profiles/r05_strict_model_ab.txt)
#endif
#ifndef ST_SPLIT_160_TRIP
#define ST_SPLIT_160_TRIP 0.8
#endif
template <typename T, bool CONV>
static int gemm_dispatch(const GemmArgs& a_in, hipStream_t st, int depth = 0) {
    GemmArgs a = a_in;
    if (a.Ng == 0) a.Ng = a.N;                   // (a whole projection: the gate rows follow the N value rows)
    const long n_eff = (a.epi & ST_EPI_GEGLU) ? 2L * a.N : a.N;
    auto tiles = [&](int bm, int bn) { return (long)cdiv(a.M, bm) * cdiv(n_eff, bn); };
    constexpr int KB = 128 / (int)sizeof(T);
    const char* who = CONV ? "conv2d" : "linear";
    if constexpr (frag2<T>()) {
        if (a.K % KB != 0) return st_fail("%s: fp8 / split fp32 operands need K to be a multiple of %d", who, KB);
    } else if (a.K % KB != 0) {                  // ragged K: register-staged kernel (no LayerNorm partials)
        if (a.stats_chunks_out) *a.stats_chunks_out = 0;
        if (tiles(128, 128) >= 240) launch_cfg<T, 128, 128, 2, 2, CONV>(a, st);
        else if (tiles(128, 64) >= 200) launch_cfg<T, 128, 64, 2, 2, CONV>(a, st);
        else launch_cfg<T, 64, 64, 2, 2, CONV>(a, st);
        return st_check_launch(who);
    }
    if constexpr (std::is_same<T, float>::value) {      // plain fp32 operands on the exact fp32 MFMA (ragged shapes of the strict mode; its matrix work runs on split operands): one configuration
        GemmArgs b = a;
        b.stats_chunks = cdiv(a.N, 64);
        if (a.stats_chunks_out) *a.stats_chunks_out = a.row_stats ? b.stats_chunks : 0;
        launch_dma<T, 64, 64, 2, 2, 4, 1, CONV>(b, st);
        return st_check_launch(who);
    } else {
        const bool even2 = (a.K % (2 * KB) == 0);
        // 8-wave blocks (two waves per SIMD hide the LDS/DMA latencies of the K loop); the tile is
        // chosen by a small cost model fitted to MI355X measurements (tools/op_bench.py):
        // one block per CU at a time, a K step costs max(address-unit time of its DMA bytes at
        // 64 B/clk, MFMA time) + a fixed sync overhead, and a partly filled last round costs a full one.
        // Tile and K split by a small cost model in microseconds, fitted to MI355X measurements
        // (tools/op_bench.py, tools/fusek_bench.py): one block per CU at a time; a K trip costs a
        // per-tile constant (set by the L2 -> LDS fill rate of ~70 GB/s per CU more than by the MFMAs);
        // a partly filled last round costs a full one; a K split adds the in-launch combine
        // (write-through fp32 slabs: ~2 us + 0.4 us per MB of slab).
        struct Cand { int cfg, bm, bn; double trip_us; };
        static const Cand cands[] = {{CFG_128x128_W8, 128, 128, 0.53}, {CFG_64x128_W8, 64, 128, 0.34}, {CFG_128x64_W8, 128, 64, 0.32},
                                     {CFG_64x64_W8, 64, 64, 0.19}, {CFG_128x320_W8, 128, 320, 1.9}, {CFG_64x320_W8, 64, 320, 1.1},
                                     {CFG_256x128_W8, 256, 128, 0.72}, {CFG_128x80_W8, 128, 80, 0.37}, {CFG_128x160_W8, 128, 160, 0.66}};
        static const int sks[] = {1, 2, 3, 4, 6, 8};
        static const int force_sk = dev_env_int("ST_GEMM_SPLITK", -1);     // 0/1: never split
        const int nk = a.K / KB;
        const long ncols = (a.epi & ST_EPI_GEGLU) ? 2L * a.N : a.N;
        const bool can_split = a.partial && !a.ln_c && force_sk != 0 && force_sk != 1;
        int cfg = CFG_64x64_W8, sk = 1;
        double best = 1e30;
        // the model over the LDS-DMA kernel's tiles for a problem of `ncols_` tile columns' worth of W rows (this launch's, or - the
        // column split below - a remainder's): returns the predicted microseconds, optionally the configuration
        auto small_model = [&](long ncols_, int* cfg_out, int* sk_out) {
            auto tiles_ = [&](int bm, int bn) { return (long)cdiv(a.M, bm) * cdiv(ncols_, bn); };
            double best_ = 1e30;
            for (const Cand& c : cands) {
                if (frag2<T>() && c.bn == 320) continue;                     // fp8 / split fragments are 32 bytes: the 320-wide wave tiles spill
                if (is_split<T>() && c.cfg == CFG_256x128_W8) continue;      // two accumulator sets: wave tiles of at most 10 x 16 x 16 (128 x 160 as 4 x 2 waves)
                if ((a.epi & ST_EPI_GEGLU) && c.bn % 32 != 0) continue;      // GEGLU: the value / gate halves of a tile are whole accumulator tiles (BN = 80 is not)
                const long nt = tiles_(c.bm, c.bn);
                // (split operands: the 128 x 160 tile runs as 4 x 2 waves with ten accumulator tiles in each of two sets: measured
                //  0.8 us per trip - it wins the GEGLU projection of the 1280 level by whole rounds, 512 tiles against 640, 104 -> 93 us,
                //  and loses q|k|v to 128 x 128 at its fitted 16-bit constant)
                // (round 5: an isolated sweep - tools/gemm_sweep.py with ST_BENCH_DTYPE=fp32 - suggested pricing the other split-operand trips at
                //  1.8x their 16-bit constants, which moved the 1280-level projections, FF2 and q|k|v of the strict step onto the 128 x 160 tile;
                //  IN the step that was 3.6 % slower at batch 1 (tools/ab_step.py --dtype fp32: 31.99 against 30.82 ms on one box; 1.4x: 31.58;
                //  level at batch 2 and 4): the fitted 16-bit constants stand, as in round 4 - factor 1.0)
                const double trip = (is_split<T>() && c.cfg == CFG_128x160_W8 ? ST_SPLIT_160_TRIP : c.trip_us * (is_split<T>() && !CONV ? ST_SPLIT_DENSE_TRIP : 1.0)) * (CONV ? 1.6 : 1.0) * ((!CONV && sizeof(T) == 2) ? model_tweak(c.cfg) : 1.0);
                for (int k_ : sks) {
                    if (k_ > 1 && (!can_split || nk / k_ < 4 || nt > 16384)) break;
                    const double slab_mb = (double)k_ * nt * c.bm * c.bn * 4.0 / 1e6;
                    if (k_ > 1 && slab_mb * 1e6 + 65536 > (double)a.partial_bytes) break;
                    const double rounds = (double)((nt * k_ + 255) / 256);
                    // with (nearly) every CU pulling, the K tiles of a round leave L2 at ~14 TB/s together: 256 blocks of
                    // 128 x 80 need 6.8 MB per trip = 0.49 us, not the 0.37 us one of them takes among 160
                    // (tools/gemm_sweep.py: 2048 x 1280 x 5120 on that tile 43 us against 33 predicted)
                    const double in_round = (double)(nt * k_ < 256 ? nt * k_ : 256);
                    const double trip_bw = in_round * (c.bm + c.bn) * 128.0 / 14.0e6;
                    // (the per-trip constants were fitted on one-round launches; launches of several rounds run 25-45 % over them
                    //  - tools/gemm_sweep.py: 2048 x 10240 x 1280 on 256 x 128 tiles 73 us against 52 predicted - hence the factor)
                    const double cost = rounds * (cdiv(nk, k_) * (trip > trip_bw ? trip : trip_bw) + 3.0) * (rounds > 1.0 ? 1.3 : 1.0) + (k_ > 1 ? (2.1 + 0.4 * slab_mb) * ((!CONV && sizeof(T) == 2) ? ST_TW_SPLITK : 1.0) : 0.0);
                    if (cost < best_) { best_ = cost; if (cfg_out) *cfg_out = c.cfg; if (sk_out) *sk_out = k_; }
                }
            }
            return best_;
        };
        best = small_model(ncols, &cfg, &sk);
        if constexpr (!CONV && sizeof(T) <= 2) {
            // the eight-phase kernel (256 x 256 or 256 x 160 tiles): no K split, whole rounds of 256 blocks.  A K step costs
            // ~1.65 us for 256 x 256 x 64 and ~1.5 us for 256 x 160 x 64 (measured, tools/gemm8p_check.py: a phase is paced by
            // its load segment - two LDS-DMA issues per wave, the fragment reads, two barriers - more than by its 12-16 MFMAs,
            // so the narrower tile buys only 8 % per step; what it buys is whole rounds: 1024 x 10240 is 256 tiles, not 160).
            // The per-trip constants above were fitted on one-round launches and run 25-45 % optimistic once a launch
            // takes several rounds (tools/gemm_sweep.py: FF1 of the 1280-channel level 51 us predicted 35, this kernel 39
            // predicted 37; QKV at batch 4 72 us against 55), so this kernel also takes the near ties.
            const int f = forced_cfg();
            double c256 = 1e30, c160 = 1e30;
            // (e4m3: a K step is 128 k - the same bytes, fragment reads and phases as a 64-k bf16 step, twice the product)
            if (gemm8p_applies(a, 256, KB)) c256 = (double)((tiles(256, 256) + 255) / 256) * (nk * 1.65 + 4.0);
            if (gemm8p_applies(a, 160, KB)) c160 = (double)((tiles(256, 160) + 255) / 256) * (nk * 1.52 + 4.0);
#ifdef ST_DEV_CONFIGS      // the four-wave kernel is a developer build's: level with the eight-phase one on the step's shapes
            if (f == CFG_256x256_4W && gemm4w_applies(a, KB)) { gemm4w_call<T>(a, st); return st_check_launch(who); }
#endif
            // (near ties go to this kernel: 1.1 - it was 1.3 while the small-tile predictions above still lacked their
            //  several-rounds and all-CUs-pulling corrections, and then took 8192 x 1920 x 640 at 40 us against 33)
            // Round 5, column split: a launch of 256 x 256 tiles whose last round is at most half full (the GEGLU projection at batch 4:
            // 640 tiles = 2.5 rounds; at batch 2: 320 = 1.25) pays a whole round for it.  The full rounds' tile COLUMNS stay on this
            // kernel; the remaining columns are a problem of their own that the model places on smaller tiles (one more launch: the
            // two write disjoint column ranges of y; GemmArgs::Ng keeps the gate rows of a GEGLU projection where they are).
            static const bool no_colsplit = dev_env_int("ST_NO_COLSPLIT", 0) != 0;
            if (f < 0 && depth == 0 && !no_colsplit && c256 < 1e29 && !a.row_stats && !a.col_stats && !a.q8_out && !a.sp_out && !a.row_scale) {
                const long tm = a.M / 256, total = tiles(256, 256);
                const long rem = total % 256;
                if (total > 256 && rem > 0 && rem <= 128 && 256 % tm == 0) {
                    const long full_cols = (total / 256) * (256 / tm);                           // tile columns of the full rounds
                    const long rem_rows = n_eff - full_cols * 256;                                // W rows (value + gate) of the remainder
                    const double c_full = (double)(total / 256) * (nk * 1.65 + 4.0);
                    const double c_split = c_full + small_model(rem_rows, nullptr, nullptr) + 2.5;
                    const double c_whole = c256 < c160 ? c256 : c160;
                    if (c_split < 0.93 * (c_whole < best ? c_whole : best)) {
                        const int bno = (a.epi & ST_EPI_GEGLU) ? 128 : 256;
                        const long n1 = full_cols * bno;                                          // output columns of the first launch
                        typedef typename OutT<T>::type TO_;
                        GemmArgs a1 = a, a2 = a;
                        a1.N = (int)n1;
                        a1.next_w = (const char*)a.W + (size_t)n1 * a.K * sizeof(T); a1.next_bytes = (size_t)(a.N - n1) * a.K * sizeof(T); a1.next_row_lines = 0; a1.next_lead_shift = 0;      // (the remainder's weights)
                        a2.N = a.N - (int)n1;
                        a2.W = (const char*)a.W + (size_t)n1 * a.K * sizeof(T);
                        a2.C = (char*)a.C + (size_t)n1 * sizeof(TO_);
                        if (a.bias) a2.bias = (const char*)a.bias + (size_t)n1 * sizeof(TO_);
                        if (a.residual) a2.residual = (const char*)a.residual + (size_t)n1 * sizeof(TO_);
                        if (a.ln_c) { a2.ln_c = a.ln_c + n1; a2.ln_d = a.ln_d + n1; }
                        if (a.col_scale) a2.col_scale = a.col_scale + n1;
                        if (a.stats_chunks_out) *a.stats_chunks_out = 0;
                        gemm8p_launch<T, 256, 2, 4>(a1, st);
                        if (int e = st_check_launch(who)) return e;
                        return gemm_dispatch<T, CONV>(a2, st, depth + 1);
                    }
                }
            }
            const bool take256 = f == CFG_256x256_8P || (f < 0 && c256 <= c160 && c256 < ST_TW_8P * best);
            const bool take160 = f == CFG_256x160_8P || (f < 0 && c160 < c256 && c160 < ST_TW_8P * best);
            if (take256 && c256 < 1e29) { gemm8p_launch<T, 256, 2, 4>(a, st); return st_check_launch(who); }
            if (take160 && c160 < 1e29) { gemm8p_launch<T, 160, 4, 2>(a, st); return st_check_launch(who); }
        }
        if constexpr (!CONV && is_split<T>()) {
            // Strict mode on the eight-phase kernel (round 5): 256 x 160 tiles only - two accumulator sets allow wave tiles of at most
            // 64 x 80.  A 32-k K tile is 120 MFMAs per SIMD (three per product) = 1,920 cycles against ~1,850 of fill and fragment
            // reads: the two wave groups put the load side under the matrix work (2,400 cycles per K tile by stamps, 80 % matrix-pipe
            // share), which the single-phase loop of gemm_dma_kernel adds to it.  Measured (tools/strict8p_ab.py): the GEGLU projections
            // 86 against 105 us at batch 1, 294 against 364 at batch 4, 370 against 485 on the 640 level at batch 4; q|k|v at batch 4
            // level, FF2 and the batch-1 q|k|v (96 tiles) stay on the smaller tiles - the model's choices below agree with every
            // measured pair.  One round of 256 blocks per 256 tiles, no K split; the fp32 epilogue costs 11-27 thousand cycles.
            const int f = forced_cfg();
            double c160 = 1e30;
            if (gemm8p_applies(a, 160, KB)) c160 = (double)((tiles(256, 160) + 255) / 256) * (nk * 1.15 + 5.0);
            if (c160 < 1e29 && (f == CFG_256x160_8P || (f < 0 && c160 < 0.9 * best))) { gemm8p_launch<T, 160, 4, 2>(a, st); return st_check_launch(who); }
        }
        GemmArgs b = a;
#ifdef ST_DEV_CONFIGS
        {   // dev knob: override only the small-problem class (fewer than 150 tiles of 128x128)
            static const int small_cfg = dev_env_int("ST_GEMM_SMALL_CFG", -1);
            if (small_cfg >= 0 && small_cfg < CFG_COUNT && tiles(128, 128) < 150 && sk == 1) cfg = small_cfg;
        }
#endif
        {   // developer overrides: ST_GEMM_FORCE=<cfg id> (tile), ST_GEMM_FUSEK=<n> (K split with that tile)
            static const int env_fk = dev_env_int("ST_GEMM_FUSEK", -1);
            int force_fk = env_fk;
#ifdef ST_DEV_CONFIGS
            if (g_dbg_cfg >= 0) force_fk = g_dbg_fusek;
#endif
            const int f = forced_cfg();
            if (f >= 0 && f < CFG_COUNT) {
                const bool u2 = (f == CFG_64x64_S4_U2 || f == CFG_128x64_S3_U2 || f == CFG_64x128_W8_U2 || f == CFG_128x64_W8_U2 ||
                                 f == CFG_64x64_W8_U2);
                if (!u2 || even2) { cfg = f; sk = (force_fk > 1 && can_split) ? (force_fk > nk ? nk : force_fk) : 1; }
            }
        }
        if (sk > 1) {
            int bm = 128;
            if (cfg == CFG_64x64_W8 || cfg == CFG_64x128_W8 || cfg == CFG_64x320_W8 || cfg == CFG_64x80_W4) bm = 64;
            if (cfg == CFG_256x128_W8 || cfg == CFG_256x160_W8_S2) bm = 256;
            const long nt = tiles(bm, cfg_bn(cfg));
            if (nt <= 16384 && (size_t)sk * nt * bm * cfg_bn(cfg) * 4 + 65536 <= a.partial_bytes) {
                // workspace layout: 16384 arrival counters (zero between launches), then the fp32 slabs
                b.splitk = sk; b.tile_counters = (int*)a.partial; b.partial = a.partial + 16384;
            }
        }
        b.stats_chunks = cdiv(a.N, cfg_bn(cfg));
        if (a.row_stats && b.stats_chunks > a.stats_capacity) return st_fail("%s: row_stats buffer holds %d chunks, %d needed", who, a.stats_capacity, b.stats_chunks);
        if (a.stats_chunks_out) *a.stats_chunks_out = a.row_stats ? b.stats_chunks : 0;
        switch (cfg) {
            case CFG_64x64_W8: launch_dma<T, 64, 64, 4, 2, 4, 1, CONV>(b, st); break;
            case CFG_128x64_W8: launch_dma<T, 128, 64, 4, 2, 4, 1, CONV>(b, st); break;
            case CFG_64x128_W8: launch_dma<T, 64, 128, 2, 4, 4, 1, CONV>(b, st); break;
            case CFG_256x128_W8: launch_dma<T, 256, 128, 4, 2, 3, 1, CONV>(b, st); break;
            case CFG_128x320_W8: launch_dma<T, 128, 320, 4, 2, 2, 1, CONV>(b, st); break;
            case CFG_64x320_W8: launch_dma<T, 64, 320, 2, 4, 3, 1, CONV>(b, st); break;
            case CFG_128x80_W8: launch_dma<T, 128, 80, 8, 1, 4, 1, CONV>(b, st); break;
            case CFG_128x160_W8:
                // (split operands: 4 x 2 waves of 32 x 80 - ten accumulator tiles in each of the two sets - instead of 8 x 1 of 16 x 160)
                if constexpr (is_split<T>()) launch_dma<T, 128, 160, 4, 2, 4, 1, CONV>(b, st);
                else launch_dma<T, 128, 160, 8, 1, 4, 1, CONV>(b, st);
                break;
#ifdef ST_DEV_CONFIGS
            case CFG_64x80_W4: launch_dma<T, 64, 80, 4, 1, 6, 1, CONV>(b, st); break;
            case CFG_128x256_W8: launch_dma<T, 128, 256, 4, 2, 3, 1, CONV>(b, st); break;       // tile/pipeline variants kept for A/B sweeps (tools/op_bench.py with ST_GEMM_FORCE)
            case CFG_64x64_S4: launch_dma<T, 64, 64, 2, 2, 4, 1, CONV>(b, st); break;
            case CFG_64x64_S8: launch_dma<T, 64, 64, 2, 2, 8, 1, CONV>(b, st); break;
            case CFG_64x64_S4_U2: launch_dma<T, 64, 64, 2, 2, 4, 2, CONV>(b, st); break;
            case CFG_128x64_S4: launch_dma<T, 128, 64, 2, 2, 4, 1, CONV>(b, st); break;
            case CFG_128x64_S3_U2: launch_dma<T, 128, 64, 2, 2, 3, 2, CONV>(b, st); break;
            case CFG_128x128_S3: launch_dma<T, 128, 128, 2, 2, 3, 1, CONV>(b, st); break;
            case CFG_64x64_S3: launch_dma<T, 64, 64, 2, 2, 3, 1, CONV>(b, st); break;
            // four-wave blocks whose LDS ring lets TWO blocks share a CU (one block's prologue / epilogue beside the other's K loop)
            case CFG_128x128_N4_S2: launch_dma<T, 128, 128, 2, 2, 2, 1, CONV>(b, st); break;
            case CFG_128x64_N4_S3: launch_dma<T, 128, 64, 2, 2, 3, 1, CONV>(b, st); break;
            case CFG_64x128_N4_S3: launch_dma<T, 64, 128, 2, 2, 3, 1, CONV>(b, st); break;
            case CFG_256x160_W8_S2: launch_dma<T, 256, 160, 4, 2, 2, 1, CONV>(b, st); break;
            case CFG_64x128_W8_S6: launch_dma<T, 64, 128, 2, 4, 6, 1, CONV>(b, st); break;
            case CFG_128x64_W8_S6: launch_dma<T, 128, 64, 4, 2, 6, 1, CONV>(b, st); break;
            case CFG_128x128_W8_S4: launch_dma<T, 128, 128, 2, 4, 3, 1, CONV>(b, st); break;       // (now the three-stage variant)
            case CFG_64x64_W8_S8: launch_dma<T, 64, 64, 4, 2, 8, 1, CONV>(b, st); break;
            case CFG_64x128_W8_U2: launch_dma<T, 64, 128, 2, 4, 3, 2, CONV>(b, st); break;
            case CFG_128x64_W8_U2: launch_dma<T, 128, 64, 4, 2, 3, 2, CONV>(b, st); break;
            case CFG_64x64_W8_U2: launch_dma<T, 64, 64, 4, 2, 4, 2, CONV>(b, st); break;
            case CFG_256x256_W8: launch_dma<T, 256, 256, 2, 4, 2, 1, CONV>(b, st); break;
            case CFG_128x128_W8_S2: launch_dma<T, 128, 128, 2, 4, 2, 1, CONV>(b, st); break;
            case CFG_128x64_W8_S3: launch_dma<T, 128, 64, 4, 2, 3, 1, CONV>(b, st); break;
            case CFG_64x128_W8_S3: launch_dma<T, 64, 128, 2, 4, 3, 1, CONV>(b, st); break;
#endif
            default: launch_dma<T, 128, 128, 2, 4, 4, 1, CONV>(b, st); break;      // CFG_128x128_W8 (four stages: long-K shapes gain 15 %)
        }
        return st_check_launch(who);
    }
}


// `next_weights` (optional argument of the three GEMM-shaped entry points): the weight matrix the launch AFTER this one
// will read; this launch touches it (one dword per 128-byte line, spread over its blocks) so it waits in the memory-side cache.
// Bits 40-59 of the byte count non-zero: a STRIDED touch - that many 128-byte lines per row of the matrix, of which the first
// 2^(bits 60-61) are touched (the K tiles the next launch's prologue asks for: a large matrix costs the touching launch its bytes at
// HBM speed, the leading columns cost 1-4 % of that and save the same cold start).
static inline void take_hint(GemmArgs& a, const void* next_w, size_t next_bytes) {
    const size_t row_lines = (next_bytes >> 40) & 0xfffff, shift = (next_bytes >> 60) & 3;
    next_bytes &= ((size_t)1 << 40) - 1;
    a.next_row_lines = 0; a.next_lead_shift = 0;
    if (next_w && row_lines && (row_lines >> shift) >= 2 && next_bytes >= (row_lines << 7)) {
        const size_t rows = (next_bytes >> 7) / row_lines;
        a.next_row_lines = (unsigned)row_lines; a.next_lead_shift = (unsigned)shift;
        next_bytes = (rows << shift) << 7;                  // what is touched
    }
    a.next_w = next_bytes ? next_w : nullptr;
    a.next_bytes = next_w ? next_bytes : 0;
}

static inline int check_epilogue(const char* who, const GemmArgs& a) {
    ST_REQUIRE(!(a.epi & ST_EPI_BIAS) || a.bias, "%s: ST_EPI_BIAS without bias pointer", who);
    ST_REQUIRE(!(a.epi & ST_EPI_RESIDUAL) || a.residual, "%s: ST_EPI_RESIDUAL without residual pointer", who);
    ST_REQUIRE(!(a.epi & ST_EPI_ROWBIAS) || (a.rowbias && a.rows_per_batch > 0), "%s: ST_EPI_ROWBIAS needs rowbias and rows_per_batch", who);
    ST_REQUIRE(!((a.epi & ST_EPI_GEGLU) && (a.epi & ST_EPI_SILU)), "%s: GEGLU and SILU are exclusive", who);
    return 0;
}
