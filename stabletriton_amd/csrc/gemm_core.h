// (gemm_core.h: every kernel and launcher template of the GEMM-shaped operators; instantiated per element type in
//  gemm_dense_*.hip / gemm_conv_*.hip / gemm_f32.hip / gemm_fp8.hip, entry points in gemm_api.hip)
// MFMA GEMM for gfx950: y[M,N] = epilogue(A[M,K] * W[N,K]^T), used for
//   * nn.Linear (rows L of SURVEY.md 8a)              - dense A loader
//   * conv2d on NHWC as implicit GEMM (row R)         - gather A loader
// bf16 / f16 run on v_mfma_f32_16x16x32_{bf16,f16}, fp32 ("strict" parity mode) on
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain); both accumulate in fp32.
//
// Structure: block tile BM x BN, K step = one 128-byte row segment (64 bf16 /
// 32 fp32), LDS double buffer with a 16-byte-chunk XOR swizzle
// (chunk ^= row & 7: conflict-free ds_read_b128 for the 16x16 fragment maps),
// register-staged global->LDS so the next tile's loads fly under the MFMAs.
// The MFMA is issued "swapped" (W fragment as the A operand, activation
// fragment as B), so each lane ends up with 4 consecutive output columns of
// one row and the epilogue stores 8/16 contiguous bytes per lane.
#pragma once
#include "common.h"
#include "attention_core.h"
#include <stdlib.h>
#include <type_traits>

struct GemmArgs {
    const void* A; const void* W; const void* bias; const void* residual; const void* rowbias; void* C;
    int M, N, K;                 // N = output columns (with GEGLU: W has 2N rows)
    long lda, ldc, ldr;
    int rows_per_batch;
    int epi;
    // implicit-GEMM conv geometry (unused for dense)
    int Hin, Win, Cin, Hout, Wout, S, stride, pad, ups;
    int R_, korder;              // filter rows; K traversal order of the conv loop (see gemm_dma_kernel)
    // LayerNorm folded into the GEMM (st_ln_linear): W already carries gamma; ln_c[n] = sum_k W'[n][k],
    // ln_d[n] = sum_k beta[k] W[n][k] (+ bias); y = rstd_m * (acc - mean_m * c_n) + d_n; the row statistics
    // come from the GEMM that produced x (it emits per-tile partial sums of the values it stores)
    const float* ln_c; const float* ln_d; float ln_eps;
    const float* ln_stats; int ln_chunks;     // per-row (sum, sum of squares) partials written by the producer GEMM
    float* row_stats; int stats_chunks;       // producer side: emit those partials, one float2 per (row, N tile)
    int stats_capacity; int* stats_chunks_out; // host-side plumbing of the chunk count
    // GroupNorm partials of the output (consumed by st_group_norm_from_stats): per tile row of the launch and per output
    // column, (sum, sum of squares) of the values stored; col_tiles_cap = tile rows the buffer holds, *col_rows_out = rows
    // per tile row actually used (host pointer; 0 = this launch emitted nothing)
    float* col_stats; int col_tiles_cap; int* col_rows_out;
    int splitk;                  // K slices (1 = none): every slice stores an fp32 slab to `partial`; the block of a tile
    float* partial;              //   that finishes last sums the slabs in slice order and runs the epilogue (in-launch combine)
    size_t partial_bytes;
    int* tile_counters;          // one arrival counter per output tile (zero between launches)
    int panel_h;                 // tile rows per panel of the block order (see gemm_dma_kernel); >= 1
    // 1x1 conv over a channel concatenation that is never materialised (st_conv1x1_cat): input channels [0, Csplit) of a
    // pixel come from A (pixel stride Csplit), the rest from A2 (pixel stride Cin - Csplit); Csplit is a multiple of a K tile
    const void* A2; int Csplit;
    // block -> tile map, prepared on the host (fill_tile_map).  Every wave of a block used to work it out with four integer
    // divisions by launch constants, ~25 scalar instructions each on the CU's one scalar unit: with the 64-bit divisions
    // of the K slices about 400 of the ~900 instructions in front of the first MFMA (2.4 us of a 12-us launch,
    // tools/gemm_probe.py).  Now: multiply-high by magic numbers (0 = divisor 1); tm_slow keeps the divisions for sizes
    // whose products leave 32 bits.
    int tm_tiles_m, tm_per_panel, tm_last_rows, tm_slow;
    unsigned tm_mg_splitk, tm_mg_per_panel, tm_mg_rows, tm_mg_last;
    int nk_base, nk_rem;         // K stages per slice: slice s takes nk_base + (s < nk_rem), slices in order
    unsigned next_per;           // 128-byte lines of next_w per touching block (0: the kernel divides)
    // fp8 operands (st_linear_fp8): acc * row_scale[m] * col_scale[n] before anything else (col_scale has 2N entries with GEGLU)
    const float* row_scale; const float* col_scale;
    int rs_stride;               // stride of row_scale: 1 = a scale per row, 0 = one scale for the whole activation tensor
    // e4m3 copy of the output for an fp8 consumer (delayed per-tensor scaling, fp8.hip): q8[m][n] = e4m3(value * *q8_inv_scale),
    // the launch's max |value| goes to the q8_amax partial slots; C may then be NULL (only the copy is wanted)
    void* q8_out; long q8_ld; const float* q8_inv_scale; unsigned int* q8_amax;
    const void* next_w; size_t next_bytes;   // weights of the NEXT launch (host hint): touched during this epilogue
    int helper_blocks;           // > 0: that many extra blocks at the end of the grid (idle CUs) do the touching instead
    // st_ln_linear_xattn: the tile is the query block of ONE head; its epilogue runs the text-context attention on it
    const void* xa_k; const void* xa_v; long xa_ldk, xa_ldv; int xa_S, xa_T; float xa_scale_log2e;
    unsigned long long* probe;   // diagnostic builds only (-DST_PROBE): per-wave phase cycle sums
};

// Developer knobs (tile / split overrides for A/B sweeps) exist only in -DST_DEV_CONFIGS builds; the product
// library never reads the environment.
static inline int dev_env_int(const char* name, int dflt) {
#ifdef ST_DEV_CONFIGS
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

static inline bool colstats_ok(const GemmArgs& a, int bm, bool lnf);      // (defined with the launchers)

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    typedef bf16x8 Frag;
    static __device__ __forceinline__ void run(f32x4& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
};
template <> struct Mma<f16> {
    typedef f16x8 Frag;
    static __device__ __forceinline__ void run(f32x4& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    typedef f32x4 Frag;
    // lane (r, q) holds k = 4q..4q+3 of this 16-wide k group; step j multiplies
    // element j of both operands, so the k permutation is the same on both sides.
    static __device__ __forceinline__ void run(f32x4& acc, const Frag& a, const Frag& b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    }
};

// fp8 (OCP e4m3) operands: A and W are bytes in memory, accumulation is fp32, everything the epilogue touches is bf16.
// The matrix instruction is the block-scaled one, v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands: 128 k per
// instruction in twice the cycles of a bf16 16x16x32, i.e. TWICE the bf16 rate (the plain v_mfma_f32_16x16x32_fp8_fp8 runs
// at the bf16 rate).  Its per-32-element E8M0 block scales are all 2^0 here (0x7F): the scales of this path are per row /
// per output channel and applied in the epilogue.  A lane (row r, lane group q) hands over 32 bytes of its row - here the
// 16-byte chunks q and q + 4 of the 128-byte K tile; which 32 of the 128 k a lane group takes is free as long as both
// operands take the same ones (the instruction sums over all of them).
struct f8 { unsigned char v; };
typedef __attribute__((ext_vector_type(8))) int i32x8;
template <> struct Mma<f8> {
    typedef i32x8 Frag;              // 32 bytes of one row: two 16-byte chunks of the K tile
    static __device__ __forceinline__ void run(f32x4& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
};
// Split fp32 operands (ST_F32S, the strict mode's matrix operands): a value x is held as two IEEE halves,
//     x ~ hi + lo * 2^-11,   hi = f16(x),   lo = f16((x - hi) * 2^11)          (22 significant bits, csrc/split.h)
// laid out so that a 128-byte row segment still holds 32 consecutive k: bytes [0, 64) the 32 hi halves, [64, 128) the 32 lo
// halves - every address computation of the fp32 path (4 bytes per element, K tiles of 32) holds unchanged, and a lane
// (row r, lane group q) finds the k = 8q .. 8q+7 of its row in the 16-byte chunks q (hi) and q + 4 (lo), exactly where
// the e4m3 path reads its two chunks.  A product takes three v_mfma_f32_16x16x32_f16 (hi.hi into the main accumulator,
// hi.lo and lo.hi into a correction accumulator that joins it times 2^-11 after the K loop; lo.lo ~ 2^-22 of the product
// is dropped): 3 x 16 cycles for 32 k against 8 x 32 cycles of v_mfma_f32_16x16x4_f32, with the same fp32 accumulation.
struct fsp { float raw; };
template <> struct Mma<fsp> {
    typedef i32x8 Frag;              // 32 bytes of one row: [0, 16) eight hi halves, [16, 32) the eight lo halves of the same k
    static __device__ __forceinline__ void run2(f32x4& acc, f32x4& corr, const Frag& a, const Frag& b) {
        typedef __attribute__((ext_vector_type(4))) int i32x4_;
        const f16x8 ah = __builtin_bit_cast(f16x8, (i32x4_)__builtin_shufflevector(a, a, 0, 1, 2, 3));
        const f16x8 al = __builtin_bit_cast(f16x8, (i32x4_)__builtin_shufflevector(a, a, 4, 5, 6, 7));
        const f16x8 bh = __builtin_bit_cast(f16x8, (i32x4_)__builtin_shufflevector(b, b, 0, 1, 2, 3));
        const f16x8 bl = __builtin_bit_cast(f16x8, (i32x4_)__builtin_shufflevector(b, b, 4, 5, 6, 7));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, corr, 0, 0, 0);
        corr = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, corr, 0, 0, 0);
    }
};
static constexpr float ST_SPLIT_INV = 1.0f / 2048.0f;      // weight of the lo halves (csrc/split.h: ST_SPLIT_SCALE = 2^11)
template <typename T> constexpr bool is_fp8() { return std::is_same<T, f8>::value; }
template <typename T> constexpr bool is_split() { return std::is_same<T, fsp>::value; }
template <typename T> constexpr bool frag2() { return is_fp8<T>() || is_split<T>(); }      // an MFMA operand = chunks q and q + 4 of the 128-byte row
template <typename T> struct OutT { typedef T type; };
template <> struct OutT<f8> { typedef bf16 type; };
template <> struct OutT<fsp> { typedef float type; };
template <typename T> constexpr int mfma_per_frag() { return is_split<T>() ? 3 : (sizeof(T) == 4 ? 4 : 1); }

template <typename T> struct Out4;
template <> struct Out4<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float* f) {
        bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (float)v[i];
    }
    static __device__ __forceinline__ void store(bf16* p, const float* f) {
        bf16x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (bf16)f[i];
        *reinterpret_cast<bf16x4*>(p) = v;
    }
};
template <> struct Out4<f16> {
    static __device__ __forceinline__ void load(const f16* p, float* f) {
        f16x4 v = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (float)v[i];
    }
    static __device__ __forceinline__ void store(f16* p, const float* f) {
        f16x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (f16)f[i];
        *reinterpret_cast<f16x4*>(p) = v;
    }
};
template <> struct Out4<float> {
    static __device__ __forceinline__ void load(const float* p, float* f) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = v[i];
    }
    static __device__ __forceinline__ void store(float* p, const float* f) {
        f32x4 v = {f[0], f[1], f[2], f[3]};
        *reinterpret_cast<f32x4*>(p) = v;
    }
};

#ifdef ST_PROBE
__device__ __forceinline__ unsigned long long probe_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define PROBE_DECL unsigned long long pr_t0 = 0, pr_a = 0, pr_b = 0, pr_c = 0, pr_d = 0, pr_x = 0; (void)pr_x;
#define PROBE_STAMP(var) unsigned long long var = probe_now();
#define PROBE_ADD(acc, t1, t0) acc += (t1) - (t0);
#else
#define PROBE_DECL
#define PROBE_STAMP(var)
#define PROBE_ADD(acc, t1, t0)
#endif

// The next launch's weights (the `next_weights` argument of the entry points) are touched one dword per 128-byte line, each block
// its slice, so that they sit in the memory-side cache when that launch starts (cold weights cost a GEMM 2-10 us:
// DESIGN.md section 6).  The loads are fire-and-forget: `sink` stays allocated until retire_touches(sink).
__device__ __forceinline__ void touch_next_weights(const GemmArgs& p, unsigned int& sink, bool helper = false) {
    if (!p.next_w || (p.helper_blocks > 0) != helper) return;
    const size_t lines = p.next_bytes >> 7;
    // slices: over the helper blocks (the last helper_blocks of the grid) when there are any, else over all blocks
    const size_t nsl = helper ? p.helper_blocks : gridDim.x;
    const size_t me = helper ? blockIdx.x - (gridDim.x - p.helper_blocks) : blockIdx.x;
    const size_t per = p.next_per ? (size_t)p.next_per : (lines + nsl - 1) / nsl;      // (host-prepared: a 64-bit division is ~130 scalar instructions)
    const size_t lo = me * per, hi = lo + per < lines ? lo + per : lines;
    const size_t step = blockDim.x;                  // read once: inside the loop the asm's memory clobber would force a reload (and a vmcnt(0)) per trip
    for (size_t l = lo + threadIdx.x; l < hi; l += step) {
        const char* a_ = (const char*)p.next_w + (l << 7);
        asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(a_) : "memory");
    }
}
// End of a touch destination's life: the loads are invisible to the compiler's waitcnt pass, so the register may only be
// handed back once they have returned.
__device__ __forceinline__ void retire_touches(unsigned int& sink) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)::"memory"); }

// ---- block -> tile map ---------------------------------------------------------------------------------------------
// floor(n / d) = mulhi(n, floor(2^32 / d) + 1) whenever n * d < 2^32 (the error term n * e / (d * 2^32), e <= d, stays below 1 / d)
static inline unsigned magic_u32(unsigned d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / d + 1); }
__device__ __forceinline__ int mg_div(int n, unsigned mg) { return mg ? (int)__umulhi((unsigned)n, mg) : n; }

struct TileId { int tile_m, tile_n, split, tw; };
// XCD-aware block order: blocks that share an XCD (blockIdx % 8) take consecutive tiles, so the W panel of a tile column is
// fetched into one L2, not eight.  Split-K: tile-major, so a tile's slices sit next to each other on one XCD, where the block
// that sums their slabs reads them fastest.  Tiles are ordered panel by panel (panel_h tile rows each), column-major inside
// a panel, so the eight contiguous XCD shares of that order are rectangles: with one panel an XCD owns whole tile columns
// (every XCD re-reads all of A, W is read once); with two or four panels an XCD re-reads 1/2 or 1/4 of A and W is read by 2
// or 4 XCDs.  The host picks what moves fewer bytes.
__device__ __forceinline__ TileId tile_of_block(const GemmArgs& p, int bid, int nblk) {
    const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    TileId r;
    if (p.tm_slow) {
        r.split = wg % p.splitk; r.tw = wg / p.splitk;
        const int pn = r.tw / p.tm_per_panel, rem = r.tw - pn * p.tm_per_panel;
        const int rows = min(p.panel_h, p.tm_tiles_m - pn * p.panel_h);
        r.tile_n = rem / rows;
        r.tile_m = pn * p.panel_h + (rem - r.tile_n * rows);
        return r;
    }
    r.tw = mg_div(wg, p.tm_mg_splitk);
    r.split = wg - r.tw * p.splitk;
    const int pn = mg_div(r.tw, p.tm_mg_per_panel), rem = r.tw - pn * p.tm_per_panel;
    const bool last = (pn + 1) * p.panel_h > p.tm_tiles_m;               // the short panel at the bottom
    const int rows = last ? p.tm_last_rows : p.panel_h;
    r.tile_n = mg_div(rem, last ? p.tm_mg_last : p.tm_mg_rows);
    r.tile_m = pn * p.panel_h + (rem - r.tile_n * rows);
    return r;
}
// (b.panel_h and b.splitk set; nk_stages = K stages of the whole problem)
static inline void fill_tile_map(GemmArgs& b, int tiles_m, int tiles_n, int nk_stages) {
    const int sk = b.splitk > 1 ? b.splitk : 1;
    b.splitk = sk;
    b.tm_tiles_m = tiles_m;
    b.tm_per_panel = b.panel_h * tiles_n;
    b.tm_last_rows = tiles_m % b.panel_h ? tiles_m % b.panel_h : b.panel_h;
    b.tm_mg_splitk = magic_u32((unsigned)sk);
    b.tm_mg_per_panel = magic_u32((unsigned)b.tm_per_panel);
    b.tm_mg_rows = magic_u32((unsigned)b.panel_h);
    b.tm_mg_last = magic_u32((unsigned)b.tm_last_rows);
    const unsigned long long blocks = (unsigned long long)tiles_m * tiles_n * sk;
    b.tm_slow = (blocks * (unsigned long long)(b.tm_per_panel > sk ? b.tm_per_panel : sk) >= (1ull << 32)) ? 1 : 0;
    b.nk_base = nk_stages / sk;
    b.nk_rem = nk_stages % sk;
}
// (after helper_blocks is decided; `grid` = blocks of the launch including helpers)
static inline void fill_next_per(GemmArgs& b, unsigned grid) {
    const size_t lines = b.next_bytes >> 7;
    const size_t nsl = b.helper_blocks > 0 ? (size_t)b.helper_blocks : (size_t)grid;
    b.next_per = (b.next_w && nsl) ? (unsigned)((lines + nsl - 1) / nsl) : 0u;
}

template <typename T> struct Raw4;
template <> struct Raw4<bf16> { typedef bf16x4 type; };
template <> struct Raw4<f16> { typedef f16x4 type; };
template <> struct Raw4<float> { typedef f32x4 type; };
template <typename T> __device__ __forceinline__ typename Raw4<T>::type ld_raw4(const T* p) {
    return *reinterpret_cast<const typename Raw4<T>::type*>(p);
}

// Sum over the sixteen lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: four single VALU instructions
// (quad butterflies, then row rotations by 4 and 8) instead of four LDS-crossbar permutes.  Fixed order: bit-reproducible.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));     // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));     // row_ror:8
    return v;
}

// Folded LayerNorm, y = rstd * (acc - mean * c) + d, as two explicit FMAs: every site that applies it (the three epilogue
// forms, the fused query-projection epilogue) must round identically - left to the compiler, one site contracted the
// multiply-adds and another did not, and st_ln_linear_xattn differed from st_ln_linear + st_attention in the last bit of a
// few fp16 outputs (bf16's 8 bits hid it).
__device__ __forceinline__ float ln_fold(float acc, float mean, float rstd, float c, float d) {
    return __builtin_fmaf(rstd, __builtin_fmaf(-mean, c, acc), d);
}

// ---- shared epilogue ---------------------------------------------------------------------------
// One output row m, 4 consecutive columns n..n+3: v = accumulators (value half), g = gate half (GEGLU).
// epilogue_compute4 does every load and all the arithmetic and leaves the final values in v;
// epilogue_put4 stores them.  The tile kernels run compute over ALL their tiles before the first
// store: on gfx950 vmcnt counts stores too, so a load issued after a store waits for that store's
// write acknowledgement (a microsecond under load) -- interleaved load/store tiles serialise on it.
template <typename T, bool GEGLU>
__device__ __forceinline__ void epilogue_compute4(const GemmArgs& p, int m, int n, float (&v)[4], const float (&g_in)[4],
                                                  float ln_mean = 0.f, float ln_rstd = 0.f) {
    const T* __restrict__ bias = (const T*)p.bias;
    const T* __restrict__ Rp = (const T*)p.residual;
    const T* __restrict__ RBp = (const T*)p.rowbias;
    const bool full = (n + 3 < p.N);
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = g_in[e];
    if (p.col_scale) {                                 // fp8 operands: dequantisation scales
        const float rs = p.row_scale[(size_t)m * p.rs_stride];
        for (int e = 0; e < 4 && n + e < p.N; ++e) { v[e] *= rs * p.col_scale[n + e]; if (GEGLU) g[e] *= rs * p.col_scale[p.N + n + e]; }
    }
    if (p.ln_c) {                                      // folded LayerNorm: rank-1 correction per row / column
        if (full) {
            float c4[4], d4[4];
            Out4<float>::load(p.ln_c + n, c4); Out4<float>::load(p.ln_d + n, d4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ln_fold(v[e], ln_mean, ln_rstd, c4[e], d4[e]);
            if (GEGLU) {
                Out4<float>::load(p.ln_c + p.N + n, c4); Out4<float>::load(p.ln_d + p.N + n, d4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = ln_fold(g[e], ln_mean, ln_rstd, c4[e], d4[e]);
            }
        } else {
            for (int e = 0; e < 4 && n + e < p.N; ++e) {
                v[e] = ln_fold(v[e], ln_mean, ln_rstd, p.ln_c[n + e], p.ln_d[n + e]);
                if (GEGLU) g[e] = ln_fold(g[e], ln_mean, ln_rstd, p.ln_c[p.N + n + e], p.ln_d[p.N + n + e]);
            }
        }
    }
    if (p.epi & ST_EPI_BIAS) {
        if (full) { float b4[4]; Out4<T>::load(bias + n, b4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b4[e];
        } else {
            for (int e = 0; e < 4 && n + e < p.N; ++e) v[e] += Elem<T>::to_f(bias[n + e]);
        }
    }
    if (GEGLU) {
        if (p.epi & ST_EPI_BIAS) {
            if (full) { float b4[4]; Out4<T>::load(bias + p.N + n, b4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] += b4[e];
            } else {
                for (int e = 0; e < 4 && n + e < p.N; ++e) g[e] += Elem<T>::to_f(bias[p.N + n + e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_for<T>(g[e]);
    }
    if (p.epi & ST_EPI_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
    }
    if (p.epi & ST_EPI_ROWBIAS) {
        const T* rb = RBp + (size_t)(m / p.rows_per_batch) * p.N + n;
        if (full) { float b4[4]; Out4<T>::load(rb, b4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b4[e];
        } else {
            for (int e = 0; e < 4 && n + e < p.N; ++e) v[e] += Elem<T>::to_f(rb[e]);
        }
    }
    if (p.epi & ST_EPI_RESIDUAL) {
        const T* rr = Rp + (size_t)m * p.ldr + n;
        if (full) { float b4[4]; Out4<T>::load(rr, b4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b4[e];
        } else {
            for (int e = 0; e < 4 && n + e < p.N; ++e) v[e] += Elem<T>::to_f(rr[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (n + e < p.N) ? Elem<T>::to_f(Elem<T>::from_f(v[e])) : 0.f;   // what will be stored
}

template <typename T>
__device__ __forceinline__ void epilogue_put4(const GemmArgs& p, int m, int n, const float (&v)[4]) {
    T* dst = (T*)p.C + (size_t)m * p.ldc + n;
    if (n + 3 < p.N) Out4<T>::store(dst, v);
    else for (int e = 0; e < 4 && n + e < p.N; ++e) dst[e] = Elem<T>::from_f(v[e]);
}

template <typename T, bool GEGLU>
__device__ __forceinline__ void epilogue_store4(const GemmArgs& p, int m, int n, float (&v)[4], const float (&g_in)[4],
                                                float ln_mean = 0.f, float ln_rstd = 0.f) {
    epilogue_compute4<T, GEGLU>(p, m, n, v, g_in, ln_mean, ln_rstd);
    epilogue_put4<T>(p, m, n, v);
}

// The feature set of an epilogue as bits (see staged_epilogue_impl): MODE >= 0 = exactly that set, tile inside the matrix.
enum { EPI_F_BIAS = 1, EPI_F_RES = 2, EPI_F_RB = 4, EPI_F_LN = 8, EPI_F_SILU = 16, EPI_F_SCALE = 32, EPI_F_ROWS = 64, EPI_F_COLS = 128,
       EPI_F_Q8 = 256, EPI_F_NOC = 512 };

// lane (r16, q) holds rows m = .. + r16, columns n = .. + 4q .. 4q+3 of every 16x16 tile.
template <typename T, int TM, int TN, int WTM, int WTN, bool GEGLU, int WGM_ = 0, int WGN_ = 0, bool ALIGNED_N = false, int MODE = -1>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                              int r16, int q, int split = 0, const float* row_mean = nullptr,
                                              const float* row_rstd = nullptr, char* lds_scratch = nullptr, int tile_n = 0,
                                              unsigned long long* ptimes = nullptr) {
    constexpr bool FAST = MODE >= 0;
    float rs1[TM], rs2[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { rs1[i] = 0.f; rs2[i] = 0.f; }
    constexpr int TNO = GEGLU ? TN / 2 : TN;
    constexpr int WTNO = GEGLU ? WTN / 2 : WTN;
    const bool has_q8 = FAST ? bool(MODE & EPI_F_Q8) : (p.q8_out != nullptr);
    const bool emit_rows = FAST ? bool(MODE & EPI_F_ROWS) : (p.row_stats != nullptr);
    const bool emit_cols = FAST ? bool(MODE & EPI_F_COLS) : (p.col_stats != nullptr && (p.N & 3) == 0);
    if (!FAST && !ALIGNED_N && (p.N & 3) != 0) {
        // ragged N: per-tile loads, arithmetic and element stores
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + r16;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TNO; ++j) {
                const int n = n0 + wn * WTNO + j * 16 + 4 * q;
                if (n >= p.N) continue;
                float v[4], g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[i][j][e]; if (GEGLU) g[e] = acc[i][j + (GEGLU ? TN / 2 : 0)][e]; }
                epilogue_store4<T, GEGLU>(p, m, n, v, g, row_mean ? row_mean[i] : 0.f, row_rstd ? row_rstd[i] : 0.f);
#pragma unroll
                for (int e = 0; e < 4; ++e) { rs1[i] += v[e]; rs2[i] = fmaf(v[e], v[e], rs2[i]); }
            }
        }
    } else {
        // Three passes: every load (unconditional, clamped addresses, so they all go out back to back
        // and cost ONE round trip), then the arithmetic, then nothing but stores.  On gfx950 vmcnt
        // counts stores too, so a load behind a store would also wait for that store's acknowledgement.
        typedef typename Raw4<T>::type R4;
        const bool has_bias = FAST ? bool(MODE & EPI_F_BIAS) : bool(p.epi & ST_EPI_BIAS), has_res = FAST ? bool(MODE & EPI_F_RES) : bool(p.epi & ST_EPI_RESIDUAL);
        const bool has_rb = FAST ? bool(MODE & EPI_F_RB) : bool(p.epi & ST_EPI_ROWBIAS), has_ln = FAST ? bool(MODE & EPI_F_LN) : (p.ln_c != nullptr);
        const bool do_silu = FAST ? bool(MODE & EPI_F_SILU) : bool(p.epi & ST_EPI_SILU), has_scale = FAST ? bool(MODE & EPI_F_SCALE) : (p.col_scale != nullptr);
        int ncol[TNO], mrow[TM];
        bool nok[TNO], mok[TM];
#pragma unroll
        for (int j = 0; j < TNO; ++j) { const int n = n0 + wn * WTNO + j * 16 + 4 * q; nok[j] = FAST || n < p.N; ncol[j] = nok[j] ? n : 0; }
#pragma unroll
        for (int i = 0; i < TM; ++i) { const int m = m0 + wm * WTM + i * 16 + r16; mok[i] = FAST || m < p.M; mrow[i] = mok[i] ? m : 0; }
        const T* __restrict__ bias = (const T*)p.bias;
        unsigned int touch_next = 0;                       // destination of the next-weights touches (kept live to the end)
        // wide wave tiles take the load + arithmetic passes in column chunks of JC tiles (registers)
        constexpr int JC = TM >= 8 ? 1 : (TNO <= 5 ? TNO : 5);      // (tall wave tiles: one column of tiles per pass)
        auto chunk = [&](auto jc) {
            constexpr int J0 = decltype(jc)::value;
            constexpr int NJ = (J0 + JC <= TNO) ? JC : TNO - J0;
            R4 braw[NJ] = {}, graw[NJ] = {};
            f32x4 cv[NJ] = {}, dv[NJ] = {}, cg[NJ] = {}, dg[NJ] = {};
            R4 rres[TM][NJ] = {}, rrb[TM][NJ] = {};
            f32x4 csc[NJ] = {}, gsc[NJ] = {};
            float rsc[TM] = {};
            if (has_scale) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    csc[j] = *reinterpret_cast<const f32x4*>(p.col_scale + ncol[J0 + j]);
                    if (GEGLU) gsc[j] = *reinterpret_cast<const f32x4*>(p.col_scale + p.N + ncol[J0 + j]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) rsc[i] = p.row_scale[(size_t)mrow[i] * p.rs_stride];
            }
            if (has_bias) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) { braw[j] = ld_raw4<T>(bias + ncol[J0 + j]); if (GEGLU) graw[j] = ld_raw4<T>(bias + p.N + ncol[J0 + j]); }
            }
            if (has_ln) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = ncol[J0 + j];
                    cv[j] = *reinterpret_cast<const f32x4*>(p.ln_c + n); dv[j] = *reinterpret_cast<const f32x4*>(p.ln_d + n);
                    if (GEGLU) { cg[j] = *reinterpret_cast<const f32x4*>(p.ln_c + p.N + n); dg[j] = *reinterpret_cast<const f32x4*>(p.ln_d + p.N + n); }
                }
            }
            if (has_rb) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) rrb[i][j] = ld_raw4<T>((const T*)p.rowbias + (size_t)(mrow[i] / p.rows_per_batch) * p.N + ncol[J0 + j]);
            }
            if (has_res) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) rres[i][j] = ld_raw4<T>((const T*)p.residual + (size_t)mrow[i] * p.ldr + ncol[J0 + j]);
            }
            if constexpr (J0 == 0) {
                // the next launch's weights: issued AFTER this pass's loads (so the arithmetic below does not wait for
                // them), in flight while the arithmetic and the stores run; the wave's exit waits for them
                touch_next_weights(p, touch_next);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float mean = row_mean ? row_mean[i] : 0.f, rstd = row_rstd ? row_rstd[i] : 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float v[4], g[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = acc[i][J0 + j][e]; g[e] = GEGLU ? acc[i][J0 + j + (GEGLU ? TN / 2 : 0)][e] : 0.f; }
                    if (has_scale) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] *= rsc[i] * csc[j][e]; if (GEGLU) g[e] *= rsc[i] * gsc[j][e]; }
                    }
                    if (has_ln) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = ln_fold(v[e], mean, rstd, cv[j][e], dv[j][e]);
                            if (GEGLU) g[e] = ln_fold(g[e], mean, rstd, cg[j][e], dg[j][e]);
                        }
                    }
                    if (has_bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += Elem<T>::to_f(braw[j][e]); if (GEGLU) g[e] += Elem<T>::to_f(graw[j][e]); }
                    }
                    if (GEGLU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= gelu_for<T>(g[e]);
                    }
                    if (do_silu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                    }
                    if (has_rb) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f(rrb[i][j][e]);
                    }
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f(rres[i][j][e]);
                    }
                    const bool live = mok[i] && nok[J0 + j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = live ? Elem<T>::to_f(Elem<T>::from_f(v[e])) : 0.f;      // what is stored
                        rs1[i] += v[e]; rs2[i] = fmaf(v[e], v[e], rs2[i]); acc[i][J0 + j][e] = v[e];
                    }
                }
            }
        };
        chunk(std::integral_constant<int, 0>{});
        if constexpr (JC < TNO) chunk(std::integral_constant<int, JC>{});
        if constexpr (2 * JC < TNO) chunk(std::integral_constant<int, 2 * JC>{});
        if constexpr (3 * JC < TNO) chunk(std::integral_constant<int, 3 * JC>{});
        static_assert(4 * JC >= TNO, "epilogue chunking covers at most four chunks");
#ifdef ST_PROBE
        if (ptimes) ptimes[0] = probe_now();
#endif
        const float q8_inv = has_q8 ? *p.q8_inv_scale : 0.f;
        float q8_max = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNO; ++j)
                if (mok[i] && nok[j]) {
                    const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    Out4<T>::store((T*)p.C + (size_t)mrow[i] * p.ldc + ncol[j], v);
                    if (has_q8) {            // e4m3 copy of the stored values (4 bytes per lane)
                        q8_max = fmaxf(fmaxf(q8_max, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                        *reinterpret_cast<unsigned int*>((unsigned char*)p.q8_out + (size_t)mrow[i] * p.q8_ld + ncol[j]) =
                            pack4_fp8(clamp_fp8(v[0] * q8_inv), clamp_fp8(v[1] * q8_inv), clamp_fp8(v[2] * q8_inv), clamp_fp8(v[3] * q8_inv));
                    }
                }
        if (has_q8) publish_amax(p.q8_amax, q8_max, blockIdx.x * 8 + (threadIdx.x >> 6));
        retire_touches(touch_next);
    }
#ifdef ST_PROBE
    if (ptimes) ptimes[1] = probe_now();
#endif
    if constexpr (WGN_ > 0) {
        // LayerNorm partials of the rows this block just stored (consumed by the next st_ln_linear):
        // lane sums -> the four q lanes -> the WGN waves of this tile row (through LDS) -> one float2
        // per (row, N tile).  Fixed order throughout: bit-reproducible.
        if (emit_rows) {
            float2* sm = reinterpret_cast<float2*>(lds_scratch);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float a1 = rs1[i], a2 = rs2[i];
                a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64);
                a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
                if (q == 0) sm[(wm * WTM + i * 16 + r16) * WGN_ + wn] = make_float2(a1, a2);
            }
            __syncthreads();
            for (int row = threadIdx.x; row < WGM_ * WTM; row += WGM_ * WGN_ * 64) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int w = 0; w < WGN_; ++w) { const float2 t = sm[row * WGN_ + w]; a1 += t.x; a2 += t.y; }
                if (m0 + row < p.M)
                    reinterpret_cast<float2*>(p.row_stats)[(size_t)(m0 + row) * p.stats_chunks + tile_n] = make_float2(a1, a2);
            }
        }
        // GroupNorm partials: per output column of this tile, (sum, sum of squares) over the tile's rows of the values just
        // stored (acc holds them, zero for rows / columns outside the problem).  In-lane over the row tiles, a fixed
        // butterfly over the sixteen row lanes, then the WGM waves of the column through LDS: bit-reproducible.
        if (emit_cols) {
            constexpr int TNO_ = GEGLU ? TN / 2 : TN;
            constexpr int WTNO_ = GEGLU ? WTN / 2 : WTN;
            float2* cm = reinterpret_cast<float2*>(lds_scratch + WGM_ * WTM * WGN_ * 8);      // behind the row-statistics area
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TNO_; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float c1 = 0.f, c2 = 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i) { const float v = acc[i][j][e]; c1 += v; c2 = fmaf(v, v, c2); }
                    c1 = row16_sum(c1); c2 = row16_sum(c2);
                    if (r16 == 0) cm[wm * (WGN_ * WTNO_) + wn * WTNO_ + j * 16 + 4 * q + e] = make_float2(c1, c2);
                }
            __syncthreads();
            const int tile_m = m0 / (WGM_ * WTM);
            for (int col = threadIdx.x; col < WGN_ * WTNO_; col += WGM_ * WGN_ * 64) {
                float c1 = 0.f, c2 = 0.f;
#pragma unroll
                for (int w = 0; w < WGM_; ++w) { const float2 t = cm[w * (WGN_ * WTNO_) + col]; c1 += t.x; c2 += t.y; }
                if (n0 + col < p.N) reinterpret_cast<float2*>(p.col_stats)[(size_t)tile_m * p.N + n0 + col] = make_float2(c1, c2);
            }
        }
    }
}


// =============================================================================
// Staged epilogue (the LDS-DMA kernels, the 256 x 256 kernel, the halo conv).  After the K loop the LDS ring is free:
// the block parks its fp32 accumulators there as a row-major tile, and every thread then owns ONE 16-byte output vector
// (8 bf16 / f16 or 4 fp32 columns of one row) per pass:
//   * stores, residual and row-bias loads are full 16-byte accesses, 128..512 contiguous bytes per row (the fragment
//     layout gave 8 bytes per lane in 32-byte row segments);
//   * per-column operands (bias, LayerNorm c / d, fp8 column scales) are loaded once per thread (its columns never change);
//   * the loads of a chunk go out BEFORE the accumulators are parked, so their latency runs under the LDS staging;
//   * a thread holds ~40 live registers instead of every epilogue operand of a whole wave tile (the 256-wide tiles and the
//     halo conv spilled 38..116 VGPRs there);
//   * GEGLU needs no value / gate pairing inside a wave any more: W rows are staged [values | gates] and the two halves of an
//     accumulator row meet in LDS, so every tile shape can carry it.
// Tiles that do not fit the ring at once go through it in row chunks (also bounding the loads in flight per thread).
// Statistics for the consumers (LayerNorm row partials, GroupNorm column partials) are reduced through LDS in a fixed order:
// bit-reproducible.  Arithmetic order per element is the one of epilogue_compute4.
// =============================================================================
template <typename TO> struct EpiVec;           // 16 bytes of outputs / residual / bias
template <> struct EpiVec<bf16> { typedef bf16x8 type; static constexpr int N = 8; };
template <> struct EpiVec<f16> { typedef f16x8 type; static constexpr int N = 8; };
template <> struct EpiVec<float> { typedef f32x4 type; static constexpr int N = 4; };

struct ColsPlain {              // accumulator n-tile j of wave column wn -> first tile column
    int wn, wtn;
    __device__ __forceinline__ int operator()(int j) const { return wn * wtn + j * 16; }
};

template <int BM, int BN, int NT, int VEC, bool GEGLU, int LDS_BYTES>
struct EpiGeom {
    static constexpr int BNO = GEGLU ? BN / 2 : BN;              // output columns of the tile
    static_assert(BNO % VEC == 0, "tile width must be a whole number of 16-byte vectors");
    static constexpr int VPR = BNO / VEC;                        // vectors (threads) per row
    static constexpr int RPI = NT / VPR;                         // rows per pass of the block
    static constexpr int LDW = BN + 4;                           // floats per staged row (+16 B: the 16 row lanes of a fragment hit different banks)
    static constexpr int ROW_BYTES = LDW * 4 + VPR * 8;          // + one (sum, sum of squares) partial per vector (row statistics)
    // passes per chunk (bounds the residual / row-bias vectors in flight; the 256 x 256 tile still holds up to 96 accumulator
    // registers of later chunks while it works on one: two passes keep it from spilling)
    static constexpr int MAX_IT = (BM * BN >= 256 * 256) ? 2 : 4;      // (four passes on the 256 x 256 tiles: no faster, and the GEGLU ones spill)
    static constexpr int ch0 = (LDS_BYTES / ROW_BYTES) / 16 * 16;
    static constexpr int ch1 = ch0 < MAX_IT * RPI ? ch0 : (MAX_IT * RPI) / 16 * 16;
    static constexpr int ch2 = ch1 < BM ? ch1 : BM;
    static constexpr int NCH = (BM + ch2 - 1) / ch2;
    static constexpr int CH = ((BM + NCH - 1) / NCH + 15) / 16 * 16;      // rows per chunk (balanced, multiple of 16)
    static constexpr int IT = (CH + RPI - 1) / RPI;
    static_assert(ch2 >= 16 && CH * ROW_BYTES <= LDS_BYTES, "staged epilogue: the ring cannot hold sixteen rows of the tile");
    static_assert(RPI * BNO * 8 <= LDS_BYTES, "staged epilogue: column-statistics scratch");
};

// What the epilogue of a launch has to do, as bits: MODE >= 0 instantiates staged_epilogue_impl for exactly that set with the
// tile known to lie inside the matrix and every pointer / stride 16-byte aligned (no per-element tests, no wide / narrow
// branches, no code for the absent features); MODE = -1 is the general instance that reads the set from the arguments.
// Why: with run-time flags the bias-only epilogue of a 256 x 256 tile took 24,000 cycles, the bare accumulators -> LDS ->
// 16-byte stores round trip 9,400 (tools/gemm_probe.py): twelve microseconds of a 48-us launch went into testing flags.

template <typename TO, int BM, int BN, int WGM, int WGN, int TM, int TN, bool GEGLU, int LDS_BYTES, bool STATS, int MODE, typename ColMap>
__device__ __forceinline__ void staged_epilogue_impl(const GemmArgs& p, f32x4 (&acc)[TM][TN], int m0, int n0, int tile_n, int wm, int r16, int q,
                                                     ColMap colmap, char* lds, const float2* lnrows, unsigned long long* ptimes = nullptr) {
    (void)ptimes;
    constexpr bool FAST = MODE >= 0;
    constexpr int NT = WGM * WGN * 64;
    constexpr int VEC = EpiVec<TO>::N;
    typedef typename EpiVec<TO>::type OV;
    typedef EpiGeom<BM, BN, NT, VEC, GEGLU, LDS_BYTES> G;
    constexpr int BNO = G::BNO, VPR = G::VPR, RPI = G::RPI, LDW = G::LDW, CH = G::CH, NCH = G::NCH, IT = G::IT;
    constexpr int WTM = BM / WGM;
    const int t = threadIdx.x;
#ifdef ST_PROBE
    if (ptimes) ptimes[2] = probe_now();
#endif
    const bool worker = t < RPI * VPR;
    const int rloc = t / VPR, v = t - rloc * VPR;
    const int n = n0 + v * VEC;                                   // first output column of this thread
#ifdef ST_EPI_MIN       // timing experiment: the bare round trip accumulators -> LDS -> 16-byte stores
    constexpr bool has_bias = false, has_res = false, has_rb = false, has_ln = false, do_silu = false, has_scale = false;
#else
    const bool has_bias = FAST ? bool(MODE & EPI_F_BIAS) : bool(p.epi & ST_EPI_BIAS), has_res = FAST ? bool(MODE & EPI_F_RES) : bool(p.epi & ST_EPI_RESIDUAL);
    const bool has_rb = FAST ? bool(MODE & EPI_F_RB) : bool(p.epi & ST_EPI_ROWBIAS), has_ln = FAST ? bool(MODE & EPI_F_LN) : (p.ln_c != nullptr);
    const bool do_silu = FAST ? bool(MODE & EPI_F_SILU) : bool(p.epi & ST_EPI_SILU), has_scale = FAST ? bool(MODE & EPI_F_SCALE) : (p.col_scale != nullptr);
#endif
    const bool has_q8 = FAST ? bool(MODE & EPI_F_Q8) : (p.q8_out != nullptr), has_c = FAST ? !(MODE & EPI_F_NOC) : (p.C != nullptr);
    const bool col_full = FAST || n + VEC <= p.N;                 // all VEC columns exist
    const bool col_any = worker && (FAST || n < p.N);
    const bool wide = col_full && (p.ldc % VEC == 0) && ((uintptr_t)p.C & 15) == 0;             // 16-byte stores
    const bool wide_res = col_full && (p.ldr % VEC == 0) && ((uintptr_t)p.residual & 15) == 0;
    const bool wide_rb = col_full && (p.N % VEC == 0) && ((uintptr_t)p.rowbias & 15) == 0;
    float* tile = reinterpret_cast<float*>(lds);
    float2* rstat = reinterpret_cast<float2*>(lds + (size_t)CH * LDW * 4);
#ifdef ST_EPI_MIN
    constexpr bool emit_rows = false, emit_cols = false;
#else
    const bool emit_rows = FAST ? bool(MODE & EPI_F_ROWS) : (STATS && p.row_stats != nullptr);
    const bool emit_cols = FAST ? bool(MODE & EPI_F_COLS) : (STATS && p.col_stats != nullptr && (p.N & 3) == 0);
#endif

    // ---- per-column operands: once per thread ------------------------------------------------------
    // (kept as loaded - raw vectors - and converted where they are used: a conversion placed here would make the compiler
    //  wait for the loads in front of the first barrier instead of letting them fly under the staging)
    OV bia = OV{}, big = OV{};
    float lc[VEC], ld[VEC], lcg[VEC], ldg[VEC], cs[VEC], csg[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { lc[e] = ld[e] = lcg[e] = ldg[e] = 0.f; cs[e] = csg[e] = 1.f; }
    if (col_any) {
        const TO* __restrict__ bias = (const TO*)p.bias;
        // whole vectors whenever the columns exist and the arrays keep 16-byte alignment (N % VEC == 0 covers the gate half too)
        const bool vec_cols = FAST || (col_full && (p.N % VEC == 0) && (!has_bias || ((uintptr_t)bias & 15) == 0) &&
                                       (!has_ln || (((uintptr_t)p.ln_c | (uintptr_t)p.ln_d) & 15) == 0) && (!has_scale || ((uintptr_t)p.col_scale & 15) == 0));
        auto ldf = [&](const float* a, float (&dst)[VEC]) {       // VEC floats
#pragma unroll
            for (int e4 = 0; e4 < VEC; e4 += 4) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(a + e4);
                dst[e4] = x[0]; dst[e4 + 1] = x[1]; dst[e4 + 2] = x[2]; dst[e4 + 3] = x[3];
            }
        };
        if (vec_cols) {
            if (has_bias) {
                bia = *reinterpret_cast<const OV*>(bias + n);
                if (GEGLU) big = *reinterpret_cast<const OV*>(bias + p.N + n);
            }
            if (has_ln) { ldf(p.ln_c + n, lc); ldf(p.ln_d + n, ld); if (GEGLU) { ldf(p.ln_c + p.N + n, lcg); ldf(p.ln_d + p.N + n, ldg); } }
            if (has_scale) { ldf(p.col_scale + n, cs); if (GEGLU) ldf(p.col_scale + p.N + n, csg); }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int ne = (n + e < p.N) ? n + e : p.N - 1;       // clamped: every load unconditional
                if (has_bias) { bia[e] = bias[ne]; if (GEGLU) big[e] = bias[p.N + ne]; }
                if (has_ln) { lc[e] = p.ln_c[ne]; ld[e] = p.ln_d[ne]; if (GEGLU) { lcg[e] = p.ln_c[p.N + ne]; ldg[e] = p.ln_d[p.N + ne]; } }
                if (has_scale) { cs[e] = p.col_scale[ne]; if (GEGLU) csg[e] = p.col_scale[p.N + ne]; }
            }
        }
    }
    const float q8_inv = has_q8 ? *p.q8_inv_scale : 0.f;           // e4m3 copy for an fp8 consumer (see GemmArgs::q8_out)
    float q8_max = 0.f;
    float c1[VEC], c2[VEC];                                       // GroupNorm partials of this thread's columns over its rows
#pragma unroll
    for (int e = 0; e < VEC; ++e) { c1[e] = 0.f; c2[e] = 0.f; }
    unsigned int touch_next = 0;

    // (this barrier costs 0.6 % of a batch-1 step - measured by leaving it out - and stays: nothing else orders the other
    //  waves' last fragment reads and tail DMAs against the tile that is about to overwrite the ring)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the last fragment reads of the K loop have returned ...
    __builtin_amdgcn_s_barrier();                                 // ... in every wave: the ring may be overwritten
#ifdef ST_PROBE
    if (ptimes) ptimes[3] = probe_now();
#endif
    // (unrolled over the chunks: static parking conditions; a rolled general instance measured 20 % slower and spilled)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int row_lo = c * CH;                                // first tile row of this chunk
        // -- loads of the chunk (16 bytes per row each), in flight while the accumulators are parked
        OV res[IT], rbv[IT];
        bool rok[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int row = row_lo + rloc + k * RPI;
            const int m = m0 + row;
            rok[k] = (rloc + k * RPI < CH) && row < BM && (FAST ? worker : (col_any && m < p.M));
            const int mc = rok[k] ? m : m0;                       // clamped
            res[k] = OV{}; rbv[k] = OV{};
            if (has_res) {
                const TO* rr = (const TO*)p.residual + (size_t)mc * p.ldr + (col_any ? n : n0);
                if (FAST || wide_res) res[k] = *reinterpret_cast<const OV*>(rr);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) res[k][e] = rr[(n + e < p.N) ? e : 0];
                }
            }
            if (has_rb) {
                const TO* rb = (const TO*)p.rowbias + (size_t)(mc / p.rows_per_batch) * p.N + (col_any ? n : n0);
                if (FAST || wide_rb) rbv[k] = *reinterpret_cast<const OV*>(rb);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) rbv[k][e] = rb[(n + e < p.N) ? e : 0];
                }
            }
        }
        if (c == 0) touch_next_weights(p, touch_next);            // the next launch's weights: fire and forget until the exit
        // -- park this chunk's accumulators (wave-uniform test: a 16-row accumulator tile lies in exactly one chunk)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * WTM + i * 16;
            if (row >= row_lo && row < row_lo + CH) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<f32x4*>(tile + (size_t)(row - row_lo + r16) * LDW + colmap(j) + 4 * q) = acc[i][j];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef ST_PROBE
        if (ptimes && c == 0) ptimes[0] = probe_now();
#endif
        // -- one 16-byte output vector per thread and pass
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int rl = rloc + k * RPI;                        // row inside the chunk
            float val[VEC];
            if (rok[k]) {
                const int row = row_lo + rl, m = m0 + row;
                const float* src = tile + (size_t)rl * LDW + v * VEC;
                float g[VEC];
#pragma unroll
                for (int e4 = 0; e4 < VEC; e4 += 4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(src + e4);
                    val[e4] = a[0]; val[e4 + 1] = a[1]; val[e4 + 2] = a[2]; val[e4 + 3] = a[3];
                    if (GEGLU) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(src + BNO + e4);
                        g[e4] = b[0]; g[e4 + 1] = b[1]; g[e4 + 2] = b[2]; g[e4 + 3] = b[3];
                    }
                }
                if (has_scale) {
                    const float rs = p.row_scale[(size_t)m * p.rs_stride];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { val[e] *= rs * cs[e]; if (GEGLU) g[e] *= rs * csg[e]; }
                }
                if (has_ln) {
                    const float2 st = lnrows[row];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        val[e] = ln_fold(val[e], st.x, st.y, lc[e], ld[e]);
                        if (GEGLU) g[e] = ln_fold(g[e], st.x, st.y, lcg[e], ldg[e]);
                    }
                }
                if (has_bias) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { val[e] += Elem<TO>::to_f(bia[e]); if (GEGLU) g[e] += Elem<TO>::to_f(big[e]); }
                }
                if (GEGLU) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) val[e] *= gelu_for<TO>(g[e]);
                }
                if (do_silu) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) val[e] = silu_f(val[e]);
                }
                if (has_rb) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) val[e] += Elem<TO>::to_f(rbv[k][e]);
                }
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) val[e] += Elem<TO>::to_f(res[k][e]);
                }
                OV out;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < VEC; ++e) out[e] = Elem<TO>::from_f(val[e]);
                if (emit_rows || emit_cols) {                     // (block-uniform: five VALU instructions per element that most launches skip)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float w = (FAST || n + e < p.N) ? Elem<TO>::to_f(out[e]) : 0.f;      // what is stored
                        s1 += w; s2 = fmaf(w, w, s2);
                        c1[e] += w; c2[e] = fmaf(w, w, c2[e]);
                    }
                }
                if (has_c) {
                    TO* dst = (TO*)p.C + (size_t)m * p.ldc + n;
                    if (FAST || wide) *reinterpret_cast<OV*>(dst) = out;
                    else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) if (n + e < p.N) dst[e] = out[e];
                    }
                }
                if constexpr (VEC == 8) {
#ifdef ST_EPI_MIN
                    if (false) {
#else
                    if (has_q8 && col_full) {
#endif        // eight e4m3 bytes per thread, 64..256 contiguous bytes per row
                        float a = 0.f;
                        unsigned int w2[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float x0 = Elem<TO>::to_f(out[4 * h]), x1 = Elem<TO>::to_f(out[4 * h + 1]), x2 = Elem<TO>::to_f(out[4 * h + 2]),
                                        x3 = Elem<TO>::to_f(out[4 * h + 3]);
                            a = fmaxf(fmaxf(a, fmaxf(fabsf(x0), fabsf(x1))), fmaxf(fabsf(x2), fabsf(x3)));
                            w2[h] = pack4_fp8(clamp_fp8(x0 * q8_inv), clamp_fp8(x1 * q8_inv), clamp_fp8(x2 * q8_inv), clamp_fp8(x3 * q8_inv));
                        }
                        q8_max = fmaxf(q8_max, a);
                        *reinterpret_cast<u32x2*>((unsigned char*)p.q8_out + (size_t)m * p.q8_ld + n) = u32x2{w2[0], w2[1]};
                    }
                }
                if (emit_rows) rstat[rl * VPR + v] = make_float2(s1, s2);
            } else if (emit_rows && worker && rl < CH) {
                rstat[rl * VPR + v] = make_float2(0.f, 0.f);
            }
        }
#ifdef ST_PROBE
        if (ptimes && c == 0) ptimes[4] = probe_now();
#endif
        if (emit_rows) {
            // LayerNorm partials of the rows just stored: one float2 per (row, N tile), the row's vectors added in order
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int rl = t; rl < CH; rl += NT) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int w = 0; w < VPR; ++w) { const float2 x = rstat[rl * VPR + w]; a1 += x.x; a2 += x.y; }
                const int row = row_lo + rl;
                if (row < BM && m0 + row < p.M)
                    reinterpret_cast<float2*>(p.row_stats)[(size_t)(m0 + row) * p.stats_chunks + tile_n] = make_float2(a1, a2);
            }
        }
        if (c + 1 < NCH || emit_cols) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this chunk's LDS reads are done before the next one is parked
            __builtin_amdgcn_s_barrier();
        }
#ifdef ST_PROBE
        if (ptimes && c == 0) ptimes[5] = probe_now();
#endif
    }
    if (emit_cols) {
        // GroupNorm partials: per output column (sum, sum of squares) over the tile's rows - the RPI row slots through LDS,
        // added in slot order
        float2* cstat = reinterpret_cast<float2*>(lds);
        if (worker) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) cstat[rloc * BNO + v * VEC + e] = make_float2(c1[e], c2[e]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int tile_m = m0 / BM;
        for (int col = t; col < BNO; col += NT) {
            float a1 = 0.f, a2 = 0.f;
            for (int w = 0; w < RPI; ++w) { const float2 x = cstat[w * BNO + col]; a1 += x.x; a2 += x.y; }
            if (n0 + col < p.N) reinterpret_cast<float2*>(p.col_stats)[(size_t)tile_m * p.N + n0 + col] = make_float2(a1, a2);
        }
    }
    if (has_q8) publish_amax(p.q8_amax, q8_max, blockIdx.x * (NT / 64) + (threadIdx.x >> 6));
    retire_touches(touch_next);
#ifdef ST_PROBE
    if (ptimes) ptimes[1] = probe_now();
#endif
}

// The dispatcher: the feature set of the launch (block-uniform), and whether this tile qualifies for a specialised instance.
// Listed are the sets the big launches of the denoise step use; anything else (and every ragged or unaligned tile) takes the
// general instance.  SCALED: e4m3 operands (row / column scales in the epilogue).
template <typename TO, int BM, int BN, int WGM, int WGN, int TM, int TN, bool GEGLU, int LDS_BYTES, bool STATS, bool SCALED = false, typename ColMap>
__device__ __forceinline__ void staged_epilogue(const GemmArgs& p, f32x4 (&acc)[TM][TN], int m0, int n0, int tile_n, int wm, int r16, int q,
                                                ColMap colmap, char* lds, const float2* lnrows, unsigned long long* ptimes = nullptr) {
#if !defined(ST_EPI_GENERIC_ONLY)
    constexpr int VEC = EpiVec<TO>::N;
    constexpr int BNO = GEGLU ? BN / 2 : BN;
    const bool has_res = p.epi & ST_EPI_RESIDUAL, has_rb = p.epi & ST_EPI_ROWBIAS;
    const bool aligned = (m0 + BM <= p.M) && (n0 + BNO <= p.N) && (p.N % VEC == 0) &&
                         (p.C == nullptr || ((p.ldc % VEC == 0) && ((uintptr_t)p.C & 15) == 0)) &&
                         (!has_res || ((p.ldr % VEC == 0) && ((uintptr_t)p.residual & 15) == 0)) && (!has_rb || ((uintptr_t)p.rowbias & 15) == 0) &&
                         ((((uintptr_t)p.bias | (uintptr_t)p.ln_c | (uintptr_t)p.ln_d | (uintptr_t)p.col_scale) & 15) == 0) &&
                         (p.q8_out == nullptr || VEC == 8);
    if (aligned) {
        const int flags = ((p.epi & ST_EPI_BIAS) ? EPI_F_BIAS : 0) | (has_res ? EPI_F_RES : 0) | (has_rb ? EPI_F_RB : 0) | (p.ln_c ? EPI_F_LN : 0) |
                          ((p.epi & ST_EPI_SILU) ? EPI_F_SILU : 0) | (p.col_scale ? EPI_F_SCALE : 0) | ((STATS && p.row_stats) ? EPI_F_ROWS : 0) |
                          ((STATS && p.col_stats && (p.N & 3) == 0) ? EPI_F_COLS : 0) | (p.q8_out ? EPI_F_Q8 : 0) | (p.C ? 0 : EPI_F_NOC);
#define ST_EPI_CASE(M)                                                                                                                          \
    case (M):                                                                                                                                   \
        staged_epilogue_impl<TO, BM, BN, WGM, WGN, TM, TN, GEGLU, LDS_BYTES, STATS, (M)>(p, acc, m0, n0, tile_n, wm, r16, q, colmap, lds, lnrows, ptimes); \
        return;
        // (a folded LayerNorm carries the projection's bias in its d vector: no BIAS bit)
        if constexpr (!STATS && !SCALED) {
            switch (flags) { ST_EPI_CASE(EPI_F_LN) ST_EPI_CASE(EPI_F_LN | EPI_F_BIAS) default: break; }
        } else if constexpr (!STATS && SCALED) {
            switch (flags) {
                ST_EPI_CASE(EPI_F_LN | EPI_F_SCALE)
                ST_EPI_CASE(EPI_F_LN | EPI_F_SCALE | EPI_F_Q8 | EPI_F_NOC)
                default: break;
            }
        } else if constexpr (STATS && !SCALED) {
            switch (flags) {
                ST_EPI_CASE(EPI_F_BIAS)
                ST_EPI_CASE(EPI_F_BIAS | EPI_F_RES)
                ST_EPI_CASE(EPI_F_BIAS | EPI_F_ROWS)
                ST_EPI_CASE(EPI_F_BIAS | EPI_F_RES | EPI_F_ROWS)
                ST_EPI_CASE(EPI_F_BIAS | EPI_F_RES | EPI_F_ROWS | EPI_F_Q8)
                ST_EPI_CASE(EPI_F_BIAS | EPI_F_COLS)
                ST_EPI_CASE(EPI_F_BIAS | EPI_F_RB | EPI_F_COLS)
                ST_EPI_CASE(EPI_F_BIAS | EPI_F_RES | EPI_F_COLS)
                default: break;
            }
        } else {
            switch (flags) {
                ST_EPI_CASE(EPI_F_BIAS | EPI_F_SCALE | EPI_F_RES | EPI_F_ROWS)
                ST_EPI_CASE(EPI_F_BIAS | EPI_F_SCALE | EPI_F_RES | EPI_F_ROWS | EPI_F_Q8)
                default: break;
            }
        }
#undef ST_EPI_CASE
    }
#endif
    staged_epilogue_impl<TO, BM, BN, WGM, WGN, TM, TN, GEGLU, LDS_BYTES, STATS, -1>(p, acc, m0, n0, tile_n, wm, r16, q, colmap, lds, lnrows, ptimes);
}

template <typename T, int BM, int BN, int WGM, int WGN, bool CONV, bool GEGLU>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_kernel(const GemmArgs p) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KB = 8 * VEC;                     // elements per 128-byte row segment
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_IT = (BM * 8 + NT - 1) / NT, B_IT = (BN * 8 + NT - 1) / NT;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static_assert(!GEGLU || (TN % 2 == 0), "GEGLU pairs value/gate n-tiles inside one wave");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile_n = blockIdx.x / tiles_m, tile_m = blockIdx.x - tile_n * tiles_m;
    const int m0 = tile_m * BM;
    constexpr int BNO = GEGLU ? BN / 2 : BN;        // output columns per block
    const int n0 = tile_n * BNO;

    const T* __restrict__ Ap = (const T*)p.A;
    const T* __restrict__ Wp = (const T*)p.W;

    // ---- per-thread staging slots: fixed (row, chunk) for the whole K loop ----
    const T* a_ptr[A_IT];      // dense: row base + chunk offset.  conv: image base + chunk offset
    int a_iy[A_IT], a_ix[A_IT];
    int a_lds[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int id = t + i * NT;
        const int row = id >> 3, c = id & 7;
        const int m = m0 + row;
        a_ok[i] = (id < BM * 8) && (m < p.M);
        a_lds[i] = row * 128 + ((c ^ (row & 7)) << 4);
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int mm = a_ok[i] ? m : 0;
            const int img = mm / hw, rem = mm - img * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_iy[i] = oy * p.stride - p.pad;
            a_ix[i] = ox * p.stride - p.pad;
            a_ptr[i] = Ap + (size_t)img * p.Hin * p.Win * p.Cin + c * VEC;
        } else {
            a_iy[i] = c * VEC;          // k offset of this chunk inside the K step
            a_ix[i] = 0;
            a_ptr[i] = Ap + (size_t)(a_ok[i] ? m : 0) * p.lda + c * VEC;
        }
    }
    const T* b_ptr[B_IT];
    int b_lds[B_IT], b_k[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int id = t + i * NT;
        const int row = id >> 3, c = id & 7;
        int wrow;                         // row of W feeding LDS row `row`
        bool ok = id < BN * 8;
        if (GEGLU) {
            const int w_ = row / WTN, local = row - w_ * WTN;
            const int half = local >= WTN / 2 ? 1 : 0;
            const int ncol = n0 + w_ * (WTN / 2) + (local - half * (WTN / 2));
            ok = ok && ncol < p.N;
            wrow = ncol + half * p.N;
        } else {
            wrow = n0 + row;
            ok = ok && wrow < p.N;
        }
        b_ok[i] = ok;
        b_k[i] = c * VEC;
        b_lds[i] = A_BYTES + row * 128 + ((c ^ (row & 7)) << 4);
        b_ptr[i] = Wp + (size_t)(ok ? wrow : 0) * p.K + c * VEC;
    }

    u32x4 a_reg[A_IT], b_reg[B_IT];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_tile = [&](int kt) {
        const int k0 = kt * KB;
        if (CONV) {
            const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
            const int r = tap / p.S, s = tap - r * p.S;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int iy = a_iy[i] + r, ix = a_ix[i] + s;
                bool ok;
                if (p.ups) {
                    ok = a_ok[i] && iy >= 0 && ix >= 0 && iy < 2 * p.Hin && ix < 2 * p.Win;
                    iy >>= 1; ix >>= 1;
                } else {
                    ok = a_ok[i] && iy >= 0 && ix >= 0 && iy < p.Hin && ix < p.Win;
                }
                const T* src = a_ptr[i] + ((size_t)(ok ? iy : 0) * p.Win + (ok ? ix : 0)) * p.Cin + c0;
                a_reg[i] = ok ? *reinterpret_cast<const u32x4*>(src) : zero4;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const bool ok = a_ok[i] && (k0 + a_iy[i] < p.K);
                a_reg[i] = ok ? *reinterpret_cast<const u32x4*>(a_ptr[i] + k0) : zero4;
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const bool ok = b_ok[i] && (k0 + b_k[i] < p.K);
            b_reg[i] = ok ? *reinterpret_cast<const u32x4*>(b_ptr[i] + k0) : zero4;
        }
    };
    auto store_tile = [&](int buf) {
        char* base = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (t + i * NT < BM * 8) *reinterpret_cast<u32x4*>(base + a_lds[i]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (t + i * NT < BN * 8) *reinterpret_cast<u32x4*>(base + b_lds[i]) = b_reg[i];
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, q = lane >> 4;
    const int nk = (p.K + KB - 1) / KB;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const char* sa = lds + cur * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = 4 * kk + q;
            Frag fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WTM + i * 16 + r16;
                fa[i] = *reinterpret_cast<const Frag*>(sa + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WTN + j * 16 + r16;
                fb[j] = *reinterpret_cast<const Frag*>(sb + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], fb[j], fa[i]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue<T, TM, TN, WTM, WTN, GEGLU>(p, acc, m0, n0, wm, wn, r16, q);
}

// =============================================================================
// v2: LDS-DMA multi-stage pipeline.  global_load_lds_dwordx4 writes each wave's
// 1 KiB (8 rows x 128 B) straight into LDS; the XOR swizzle is applied on the
// per-lane SOURCE address (the LDS destination of an LDS-DMA is lane-linear),
// STAGES buffers keep STAGES-1 K-tiles in flight behind a counted vmcnt and a raw
// s_barrier (a __syncthreads() would drain the DMA queue).  Rows outside M / N
// and padded conv taps read from a 16-byte zero buffer, so no lane is masked.
// =============================================================================
static __device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};      // (one per translation unit)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

#ifndef ST_AUX_A
#define ST_AUX_A 0
#endif
#ifndef ST_AUX_B
#define ST_AUX_B 0
#endif
// cache-policy bits of the DMA (aux: 1 = sc0, 2 = nt, 16 = sc1): default policy for both operands - every
// tile is re-read by the other blocks of its tile row / column through the XCD's L2 (measured: nt on
// either stream is slower)
template <int AUX = 0>
__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)lds_wave_base, 16, 0, AUX);
}

// number of a stage's G DMA entries that the first NG-1 MFMA groups issue (entry e goes with group e*NG/G)
constexpr int dma_before_last_group(int G, int NG) {
    int n = 0;
    for (int e = 0; e < G; ++e) n += (e * NG / G < NG - 1) ? 1 : 0;
    return n;
}

constexpr int dma_in_group(int G, int NG, int g) {
    int n = 0;
    for (int e = 0; e < G; ++e) n += (e * NG / G == g) ? 1 : 0;
    return n;
}

// The same with the destination as an LDS byte address (what an address_space(3) pointer is).  For destinations picked by
// a select (live piece or dump area): the generic -> LDS conversion of a selected pointer carries a null test, and on one
// instantiation (128 x 160, GEGLU, LayerNorm fold) hipcc 7.2 emitted "V_CMP_NE_U32 0, src_shared_base" for it and
// stopped with "Illegal instruction detected".
template <int AUX = 0>
__device__ __forceinline__ void dma16_at(const void* src, unsigned lds_addr) {
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)(uintptr_t)lds_addr, 16, 0, AUX);
}
__device__ __forceinline__ unsigned lds_addr_of(const char* p) { return (unsigned)(uintptr_t)(lds_void_t*)p; }

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// Row statistics of a LayerNorm-folded GEMM: the producer left per row one (sum, sum of squares) partial per N tile.
// TPR adjacent threads share a row.  The order of the additions is CANONICAL - independent of TPR, i.e. of the tile
// configuration the dispatch picked: eight strided partial sums (chunk c goes to class c mod 8, added in ascending c)
// combined by the fixed tree ((0+1)+(2+3))+((4+5)+(6+7)).  A thread owns 8 / TPR classes; the tree's lower levels are a
// butterfly over the TPR threads, its upper levels run inside the thread.  Without this the same x gave statistics that
// differed in the last bit between a 64-row and a 128-row tile, and with them a few fp16 outputs.
// load(): every load unconditional with clamped indices (one round trip, issued ahead of the prologue DMA).
template <int TPR>
struct LnRowSum {
    static_assert(TPR == 1 || TPR == 2 || TPR == 4 || TPR == 8, "threads per row");
    static constexpr int RES = 8 / TPR;          // classes per thread
    static constexpr int PRE = 8 / RES;          // preloaded chunks per class (8 loads per thread in all)
    float2 pre[RES][PRE];
    __device__ __forceinline__ void load(const float2* row, int chunks, int part) {
#pragma unroll
        for (int j = 0; j < RES; ++j)
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                const int c = part + TPR * j + 8 * i;
                pre[j][i] = row[c < chunks ? c : 0];
            }
    }
    __device__ __forceinline__ void finish(const float2* row, int chunks, int part, float& s1, float& s2) {
        float a1[RES], a2[RES];
#pragma unroll
        for (int j = 0; j < RES; ++j) {
            a1[j] = 0.f; a2[j] = 0.f;
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                const bool ok = part + TPR * j + 8 * i < chunks;
                a1[j] += ok ? pre[j][i].x : 0.f; a2[j] += ok ? pre[j][i].y : 0.f;
            }
            for (int c = part + TPR * j + 8 * PRE; c < chunks; c += 8) { const float2 v = row[c]; a1[j] += v.x; a2[j] += v.y; }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) { a1[j] += __shfl_xor(a1[j], o, 64); a2[j] += __shfl_xor(a2[j], o, 64); }
        }
#pragma unroll
        for (int w = 1; w < RES; w <<= 1)
#pragma unroll
            for (int j = 0; j + w < RES; j += 2 * w) { a1[j] += a1[j + w]; a2[j] += a2[j + w]; }
        s1 = a1[0]; s2 = a2[0];
    }
};

// In-launch split-K combine (cdna guide, projection GEMM item 2).  Every slice stores its fp32
// accumulators as a slab in FRAGMENT order (a wave-instruction writes 1 KiB contiguous) with
// write-through stores and draws a ticket; the block that draws the last ticket re-reads ALL slabs in
// slice order (bit-reproducible whichever block is last) and goes on to the epilogue (returns true).
// Nobody waits on anybody, so there is no spin to hang in.  `lds` lends one word for the ticket.
template <int TM, int TN, int TILE_ELEMS>
__device__ __forceinline__ bool splitk_combine(const GemmArgs& p, f32x4 (&acc)[TM][TN], int tw, int split, char* lds, int t, int wave, int lane) {
    float* slab0 = p.partial + (size_t)tw * p.splitk * TILE_ELEMS;
    {
        // the stores below are inline asm, which the compiler's hazard pass does not protect against the MFMAs that have
        // just written `acc` (no hardware interlock either): 19 wait states cover the longest (16-pass) MFMA
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
        float* mine = slab0 + (size_t)split * TILE_ELEMS + (size_t)wave * (TM * TN * 256) + lane * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // write-through (sc1) store: visible to every XCD once acknowledged, no release fence needed
                const float* dst = mine + (i * TN + j) * 256;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(acc[i][j]) : "memory");
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its own stores ...
    __syncthreads();                                          // ... before the one lane that signals for all
    int* flag = reinterpret_cast<int*>(lds);
    if (t == 0) *flag = __hip_atomic_fetch_add(p.tile_counters + tw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != p.splitk - 1) return false;
    if (t == 0) __hip_atomic_store(p.tile_counters + tw, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    // This block's own slice is in its registers: it is added from there, at its place in the slice order (the slab holds
    // the very same fp32 values, so the sum is the one a read-back would give), and a sixth to a half of the slab reads
    // of the last arriver - which pulls them through ONE CU's load path - disappear.
    f32x4 own[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { own[i][j] = acc[i][j]; acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // Every slab load carries sc1 (served past this CU's L1, which other CUs' write-through stores never refresh), as a
    // raw buffer load so that it stays compiler-visible: the destination of an inline-asm load may be copied or spilled by
    // the compiler before the data has arrived (seen as soon as a 128-accumulator tile put the register file under
    // pressure); here the compiler counts the loads itself.
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t slabs = __builtin_amdgcn_make_buffer_rsrc((void*)slab0, 0, (int)((size_t)p.splitk * TILE_ELEMS * 4), 0x00020000);
    for (int sl = 0; sl < p.splitk; ++sl) {
        if (sl == split) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] += own[i][j];
            continue;
        }
        const int off = (sl * TILE_ELEMS + wave * (TM * TN * 256) + lane * 4) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] += __builtin_bit_cast(f32x4, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(slabs, off + (i * TN + j) * 1024, 0, 16));
    }
    return true;
}

template <typename T, int BM, int BN, int WGM, int WGN, int STAGES, int U, bool CONV, bool GEGLU, bool LNF = false, bool XA = false>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_dma_kernel(const GemmArgs p) {
    constexpr int NW = WGM * WGN;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KB = 8 * VEC;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    // 1-KiB row blocks (DMA pieces) per wave.  When the pieces of a tile do not divide over the waves
    // (BN = 80: ten pieces), the waves left without one issue a dummy DMA (16 zero bytes for every lane,
    // one cache line) into a dump area, so that every wave counts the same vmcnt.
    constexpr int A_PIECES = BM / 8, B_PIECES = BN / 8;
    constexpr int A_IT = (A_PIECES + NW - 1) / NW, B_IT = (B_PIECES + NW - 1) / NW;
    constexpr bool UNEVEN = (A_PIECES % NW != 0) || (B_PIECES % NW != 0);
    constexpr int G = (A_IT + B_IT) * U;                         // DMA instructions per wave per stage
    constexpr int A_BYTES = BM * 128, TILE = (BM + BN) * 128, STAGE = TILE * U;   // a stage = U consecutive K tiles
    static_assert(!GEGLU || (BN % 32 == 0), "GEGLU: value and gate halves of the tile are whole 16-column accumulator tiles");
    static_assert((STAGES - 2) * G <= 63, "vmcnt immediate");
    typedef typename Mma<T>::Frag Frag;
    typedef typename OutT<T>::type TO;                          // element type of C, bias, residual (fp8 operands: bf16)

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const dump = lds + STAGES * STAGE + BM * 8;          // after the ring and the LayerNorm (mean, rstd) rows

#ifdef ST_PROBE
    unsigned long long pr_k0 = probe_now(), pr_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pr_rt0)::"memory");
#endif
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int tiles_m = (p.M + BM - 1) / BM;
    // XCD-aware block order: blocks that share an XCD (blockIdx % 8) take consecutive
    // tiles, so the W panel of a tile column is fetched into one L2, not eight
    const int nblk = gridDim.x - p.helper_blocks, bid = blockIdx.x;
    if (bid >= nblk) {                               // helper block on an otherwise idle CU: the next launch's weights
        unsigned int sink = 0;
        touch_next_weights(p, sink, true);
        retire_touches(sink);
        return;
    }
    constexpr int BNO = GEGLU ? BN / 2 : BN;
    const TileId tid = tile_of_block(p, bid, nblk);
    const int split = tid.split, tw = tid.tw, tile_m = tid.tile_m, tile_n = tid.tile_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BNO;

    const T* __restrict__ Ap = (const T*)p.A;
    const T* __restrict__ Wp = (const T*)p.W;
    const T* zeros = reinterpret_cast<const T*>(g_zero16);

    // ---- per-lane DMA sources: fixed (row, logical chunk) for the whole K loop ----
    const int lr = lane >> 3;                     // row inside the 8-row block
    const int lc = (lane & 7) ^ lr;               // logical 16-byte chunk this lane fetches (source-side swizzle)
    const T* a_ptr[A_IT];
    const T* a2_ptr[CONV ? A_IT : 1];              // two-source 1x1 conv: the pixel's row in the second tensor
    int a_adv[A_IT], a_iy[A_IT], a_ix[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = (wave + i * NW) * 8 + lr;
        const int m = m0 + row;
        const bool ok = m < p.M && (!UNEVEN || wave + i * NW < A_PIECES);
        if (CONV && p.A2) {        // 1x1, stride 1: input pixel = output pixel m
            a_ptr[i] = Ap + (size_t)(ok ? m : 0) * p.Csplit + lc * VEC;
            a2_ptr[i] = (const T*)p.A2 + (size_t)(ok ? m : 0) * (p.Cin - p.Csplit) + lc * VEC;
            a_iy[i] = ok ? 0 : -(1 << 28); a_ix[i] = 0; a_adv[i] = 0;
        } else if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int mm = ok ? m : 0;
            const int img = mm / hw, rem = mm - img * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_iy[i] = ok ? oy * p.stride - p.pad : -(1 << 28);
            a_ix[i] = ox * p.stride - p.pad;
            a_ptr[i] = Ap + (size_t)img * p.Hin * p.Win * p.Cin + lc * VEC;
            a_adv[i] = 0;
        } else {
            a_ptr[i] = ok ? Ap + (size_t)m * p.lda + lc * VEC : zeros;
            a_adv[i] = ok ? KB : 0;
            a_iy[i] = a_ix[i] = 0;
        }
    }
    const T* b_ptr[B_IT];
    int b_adv[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = (wave + i * NW) * 8 + lr;
        int wrow;
        bool ok;
        if (GEGLU) {                  // LDS rows [0, BN/2) = value rows of W, [BN/2, BN) = gate rows (N rows further down)
            const int half = row >= BN / 2 ? 1 : 0;
            const int ncol = n0 + row - half * (BN / 2);
            ok = ncol < p.N && (!UNEVEN || wave + i * NW < B_PIECES);
            wrow = ncol + half * p.N;
        } else {
            wrow = n0 + row;
            ok = wrow < p.N && (!UNEVEN || wave + i * NW < B_PIECES);
        }
        b_ptr[i] = ok ? Wp + (size_t)wrow * p.K + lc * VEC : zeros;
        b_adv[i] = ok ? KB : 0;
    }

    // Touch the epilogue's operands now (one dword per 128-byte line, value unused): they are first
    // read after the K loop, where a miss to HBM would be fully exposed.  These loads are older than
    // every DMA, so the counted vmcnt waits below retire them for free.
    // The destination register stays reserved (touch_sink is "used" after the prologue wait that
    // retires the loads), so a late return cannot land in a register that has been given away.
    unsigned int touch_sink = 0;
    {
        constexpr int NT_ = NW * 64;
        const int ncols_out = min(BNO, p.N - n0);                      // output columns of this tile
        auto touch_at = [&](const char* a) {
            a = (const char*)((uintptr_t)a & ~(uintptr_t)3);
            asm volatile("global_load_dword %0, %1, off" : "+v"(touch_sink) : "v"(a) : "memory");
        };
        auto touch = [&](const void* base, long byte_off, int nbytes) {
            for (int o = t * 128; o < nbytes; o += NT_ * 128) touch_at((const char*)base + byte_off + o);
        };
        if (p.epi & ST_EPI_BIAS) {
            touch(p.bias, (long)n0 * sizeof(TO), ncols_out * (int)sizeof(TO));
            if (GEGLU) touch(p.bias, ((long)p.N + n0) * sizeof(TO), ncols_out * (int)sizeof(TO));
        }
        if (LNF) {
            touch(p.ln_c, (long)n0 * 4, ncols_out * 4); touch(p.ln_d, (long)n0 * 4, ncols_out * 4);
            if (GEGLU) { touch(p.ln_c, ((long)p.N + n0) * 4, ncols_out * 4); touch(p.ln_d, ((long)p.N + n0) * 4, ncols_out * 4); }
        }
        if (p.epi & ST_EPI_RESIDUAL) {
            const int lines = (ncols_out * (int)sizeof(TO) + 127) / 128;      // per row
            const int rows = min(BM, p.M - m0);
            for (int o = t; o < rows * lines; o += NT_) {
                const int r = o / lines, l = o - r * lines;
                touch_at((const char*)p.residual + ((size_t)(m0 + r) * p.ldr + n0) * sizeof(TO) + l * 128);
            }
        }
    }

    // K range of this block in stages (host guarantees K % (KB*U) == 0); split-K slices are balanced
    const int nk_lo = split * p.nk_base + min(split, p.nk_rem), nk_hi = nk_lo + p.nk_base + (split < p.nk_rem ? 1 : 0);
    const int kbase = nk_lo * U;
    // DMA list of a stage: for each of its U tiles, A_IT activation pieces then B_IT weight pieces.
    // `issue_range` emits entries [lo, hi) so the loop can spread them between MFMA groups
    // (back-to-back DMAs serialise in the address unit while the matrix pipe idles).
    constexpr int PER_TILE = A_IT + B_IT;
    // conv: position of the K tile that the next issue fetches, advanced once per stage (no divisions
    // in the loop; a K tile never straddles a filter tap because Cin is a multiple of the tile)
    int cs_r = 0, cs_s = 0, cs_c0 = 0;
    // p.korder 1 walks K channel-slice-major (all R*S taps of 64 channels, then the next 64 channels): the
    // nine shifted windows of one channel slice follow each other, so most of their lines are still in the
    // CU's L1 when the next tap asks for them; 0 is tap-major (the memory order of W's K axis).
    if (CONV) {
        if (p.korder) {
            const int taps = p.R_ * p.S;
            const int cs = kbase / taps, tap = kbase - cs * taps;
            cs_c0 = cs * KB; cs_r = tap / p.S; cs_s = tap - cs_r * p.S;
        } else {
            const int k0 = kbase * KB;
            const int tap = k0 / p.Cin;
            cs_c0 = k0 - tap * p.Cin; cs_r = tap / p.S; cs_s = tap - cs_r * p.S;
        }
    }
    auto conv_advance = [&](bool go) {              // branch-free: `go` false leaves the position where it is
        if (p.korder) {
            cs_s += go ? 1 : 0;
            const bool w1 = cs_s == p.S;
            cs_s = w1 ? 0 : cs_s;
            cs_r += w1 ? 1 : 0;
            const bool w2 = cs_r == p.R_;
            cs_r = w2 ? 0 : cs_r;
            cs_c0 += w2 ? KB : 0;
        } else {
            cs_c0 += go ? KB * U : 0;
            const bool w1 = cs_c0 >= p.Cin;
            cs_c0 -= w1 ? p.Cin : 0;
            cs_s += w1 ? 1 : 0;
            const bool w2 = cs_s == p.S;
            cs_s = w2 ? 0 : cs_s;
            cs_r += w2 ? 1 : 0;
        }
    };
    const unsigned lds_base = lds_addr_of(lds), dump_addr = lds_base + STAGES * STAGE + BM * 8;       // (= dump)
    auto issue_one = [&](int st, int buf, int e) {
        const int u = e / PER_TILE, i = e - u * PER_TILE;
        const int kt = kbase + st * U + u;
        const unsigned base = lds_base + buf * STAGE + u * TILE;
        if (i < A_IT) {
            const T* src;
            if (CONV) {
                int r, s_, c0;
                if constexpr (U == 1) {            // running (tap row, tap column, channel offset) of the stage being fetched
                    r = cs_r; s_ = cs_s; c0 = cs_c0;
                } else {
                    const int k0 = kt * KB;
                    const int tap = k0 / p.Cin;
                    c0 = k0 - tap * p.Cin; r = tap / p.S; s_ = tap - r * p.S;
                }
                int iy = a_iy[i] + r, ix = a_ix[i] + s_;
                bool ok;
                if (p.ups) {
                    ok = iy >= 0 && ix >= 0 && iy < 2 * p.Hin && ix < 2 * p.Win;
                    iy >>= 1; ix >>= 1;
                } else {
                    ok = iy >= 0 && ix >= 0 && iy < p.Hin && ix < p.Win;
                }
                src = ok ? a_ptr[i] + ((size_t)iy * p.Win + ix) * p.Cin + c0 : zeros;
                if (p.A2) src = a_iy[i] < 0 ? zeros : (c0 < p.Csplit ? a_ptr[i] + c0 : a2_ptr[CONV ? i : 0] + (c0 - p.Csplit));
#ifdef ST_CONV_SKIP_A
                if (r != 0 || s_ != 0) src = zeros;          // timing experiment: fetch the input for one tap in nine
#endif
            } else {
                src = a_ptr[i] + (size_t)kt * a_adv[i];
            }
            const int pa = wave + i * NW;
            dma16_at<ST_AUX_A>(src, (!UNEVEN || pa < A_PIECES) ? base + pa * 1024 : dump_addr);
        } else {
            const int j = i - A_IT;
            const int pb = wave + j * NW;
            const T* bsrc;
            if (CONV && U == 1) bsrc = b_ptr[j] + (b_adv[j] ? (size_t)((cs_r * p.S + cs_s) * p.Cin + cs_c0) : 0);     // W[n][tap][c]
            else bsrc = b_ptr[j] + (size_t)kt * b_adv[j];
            dma16_at<ST_AUX_B>(bsrc, (!UNEVEN || pb < B_PIECES) ? base + A_BYTES + pb * 1024 : dump_addr);
        }
    };
    auto issue = [&](int st, int buf) {
#pragma unroll
        for (int e = 0; e < G; ++e) issue_one(st, buf, e);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // split fp32 operands: the cross products (hi.lo + lo.hi, in units of 2^-11) accumulate apart from the main products
    f32x4 corr[is_split<T>() ? TM : 1][is_split<T>() ? TN : 1];
    if constexpr (is_split<T>()) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) corr[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int r16 = lane & 15, q = lane >> 4;
    const int nk = nk_hi - nk_lo;

    // LayerNorm-folded GEMM: the producer left per-row (sum, sum of squares) partials, one float2 per
    // (row, producer N tile).  TPR adjacent threads share a row: each loads every TPR-th partial (all
    // loads unconditional with clamped indices, so they cost one round trip, issued ahead of the
    // prologue DMA), a fixed-order butterfly adds them, and (mean, rstd) wait in LDS for the epilogue.
    constexpr int TPR = (NW * 64 / BM) >= 8 ? 8 : (NW * 64 / BM >= 1 ? NW * 64 / BM : 1);      // (threads beyond 8 per row idle here)
    constexpr int TPR_SPAN = NW * 64 / BM >= 1 ? NW * 64 / BM : 1;                                 // threads that map to one row
    LnRowSum<TPR> ln_sum;
    if constexpr (LNF) {
        static_assert(NW * 64 % BM == 0 && (TPR_SPAN & (TPR_SPAN - 1)) == 0, "threads per row must be a power of two");
        const float2* st2 = reinterpret_cast<const float2*>(p.ln_stats);
        const int row = t / TPR_SPAN, part = (t - row * TPR_SPAN) & (TPR - 1);
        const int m = min(m0 + row, p.M - 1);
        ln_sum.load(st2 + (size_t)m * p.ln_chunks, p.ln_chunks, part);
    }
#pragma unroll
    for (int s_ = 0; s_ < STAGES - 1; ++s_)
        if (s_ < nk) { issue(s_, s_); if (CONV) conv_advance(s_ < nk - 1); }
    if constexpr (LNF) {
        const float2* st2 = reinterpret_cast<const float2*>(p.ln_stats);
        const int row = t / TPR_SPAN, sub = t - row * TPR_SPAN, part = sub & (TPR - 1);
        const int m = min(m0 + row, p.M - 1);
        float a1, a2;
        ln_sum.finish(st2 + (size_t)m * p.ln_chunks, p.ln_chunks, part, a1, a2);
        const float mean = a1 / (float)p.K;
        const float rstd = rsqrtf(fmaxf(a2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
        if (sub == 0) reinterpret_cast<float2*>(lds + STAGES * STAGE)[row] = make_float2(mean, rstd);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // written before the raw barrier below
    }
    if (nk >= STAGES - 1) wait_vmcnt<(STAGES - 2) * G>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::"v"(touch_sink));            // the touch loads have returned by now

    // Software pipeline (one wave per SIMD has nobody else to hide LDS latency behind):
    // the fragments of MFMA group g+1 are read while group g multiplies, and the LAST group
    // of a stage multiplies after the stage barrier, under the first reads of the next stage.
    constexpr int GPT = frag2<T>() ? 1 : 2;        // MFMA groups per K tile: two 64-byte halves; fp8 / split fp32: the whole 128-byte row per operand
    constexpr int NG = GPT * U;                   // MFMA groups per stage
    constexpr int RPF = frag2<T>() ? 2 : 1;        // 16-byte LDS reads per fragment
    // with only two buffers the whole prefetch must be issued before the stage barrier (group 0)
    constexpr bool EARLY = (STAGES == 2);
    Frag fa[2][TM], fb[2][TN];
    auto read_frag = [&](const char* base, int row, int g) -> Frag {
        if constexpr (frag2<T>()) {
            const u32x4 lo = *reinterpret_cast<const u32x4*>(base + row * 128 + ((q ^ (row & 7)) << 4));
            const u32x4 hi = *reinterpret_cast<const u32x4*>(base + row * 128 + (((q + 4) ^ (row & 7)) << 4));
            return Frag{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        } else {
            const int c = 4 * (g & 1) + q;
            return *reinterpret_cast<const Frag*>(base + row * 128 + ((c ^ (row & 7)) << 4));
        }
    };
    auto read_group = [&](int buf, int g, int set) {
#ifdef ST_FILL_ONLY
        (void)buf; (void)g; (void)set; return;        // timing experiment: DMA stream only
#endif
        const char* sa = lds + buf * STAGE + (g / GPT) * TILE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[set][i] = read_frag(sa, wm * WTM + i * 16 + r16, g);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[set][j] = read_frag(sb, wn * WTN + j * 16 + r16, g);
    };
    auto mma_group = [&](int set) {
#if defined(ST_FILL_ONLY)
        (void)set; return;                              // timing experiment: no matrix work
#elif defined(ST_NO_MMA)
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[set][i]));      // keep the fragment reads alive
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[set][j]));
        return;
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (is_split<T>()) Mma<T>::run2(acc[i][j], corr[i][j], fb[set][j], fa[set][i]);
                else Mma<T>::run(acc[i][j], fb[set][j], fa[set][i]);
            }
    };

    int cur = 0, nxt = STAGES - 1;
    PROBE_DECL
    PROBE_STAMP(pr_start)
    read_group(0, 0, 0);
    // The trip body is branch-free: trips past the last prefetch re-fetch the final stage into a
    // buffer nobody reads again, so the vmcnt bookkeeping is the same every trip.
    // PAR: which of the two fragment register sets group 0 of this trip multiplies from - with an odd number of groups
    // per stage (fp8: one) the sets trade places every trip, so the loop runs two trips per iteration.
    auto trip = [&](int kt, auto par_) {
        constexpr int PAR = decltype(par_)::value;
        PROBE_STAMP(pr_i0)
        const int pf = min(kt + STAGES - 1, nk - 1);      // stage to prefetch (clamped)
        auto group = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 1 < NG) {
                read_group(cur, g + 1, (g + 1 + PAR) & 1);
            } else {
                // stage kt+1 must have landed (own DMAs), then everyone's; the barrier also retires
                // every wave's reads of `cur` (all of them are in registers by now) before its refill
                PROBE_STAMP(pr_i1)
                // in flight at this point: stages kt+2 .. kt+S-2 whole, plus the shares of stage
                // kt+S-1 already issued by groups 0 .. NG-2 of this trip
                wait_vmcnt<EARLY ? 0 : (STAGES - 3) * G + dma_before_last_group(G, NG)>();
                PROBE_STAMP(pr_i2)
                __builtin_amdgcn_s_barrier();
                PROBE_STAMP(pr_i3)
                PROBE_ADD(pr_a, pr_i1, pr_i0) PROBE_ADD(pr_b, pr_i2, pr_i1) PROBE_ADD(pr_c, pr_i3, pr_i2)
                __builtin_amdgcn_sched_barrier(0);
                read_group(cur + 1 == STAGES ? 0 : cur + 1, 0, (g + 1 + PAR) & 1);
            }
            constexpr int n_dma = EARLY ? (g == 0 ? G : 0) : dma_in_group(G, NG, g);
#pragma unroll
            for (int e = 0; e < G; ++e)
                if ((EARLY ? 0 : e * NG / G) == g) issue_one(pf, nxt, e);
            mma_group((g + PAR) & 1);
            // pin the emitted order of this group: fragment reads of the NEXT group first, then this
            // group's DMA share, then this group's MFMAs (hipcc otherwise sinks the reads to just
            // before their use and exposes the LDS latency in front of every MFMA cluster)
#ifndef ST_NO_PIN
            __builtin_amdgcn_sched_group_barrier(0x100, (TM + TN) * RPF, 0);
            if constexpr (n_dma > 0) __builtin_amdgcn_sched_group_barrier(0x020, n_dma, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN * mfma_per_frag<T>(), 0);
            __builtin_amdgcn_sched_barrier(0);
#else
            (void)n_dma;
#endif
        };
        group(std::integral_constant<int, 0>{});
        if constexpr (NG > 1) group(std::integral_constant<int, 1>{});
        if constexpr (NG > 2) {
            group(std::integral_constant<int, 2>{});
            group(std::integral_constant<int, 3>{});
        }
        // the prefetched first fragments of the next stage have had a whole MFMA group to land: retire
        // them here so the compiler enters the next trip with an empty LDS scoreboard (exact waits)
#ifndef ST_NO_PIN
        __builtin_amdgcn_s_waitcnt(0xc07f);
#endif
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
        if (CONV) conv_advance(kt + STAGES - 1 < nk - 1);
    };
    if constexpr (NG % 2 == 0) {
        for (int kt = 0; kt < nk; ++kt) trip(kt, std::integral_constant<int, 0>{});
    } else {
        for (int kt = 0; kt < nk; kt += 2) {
            trip(kt, std::integral_constant<int, 0>{});
            if (kt + 1 < nk) trip(kt + 1, std::integral_constant<int, 1>{});
        }
    }
    wait_vmcnt<0>();                              // no LDS-DMA may outlive the workgroup's LDS allocation
    if constexpr (is_split<T>()) {                // one fused multiply-add per element: the same bits whatever follows
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = __builtin_fmaf(corr[i][j][e], ST_SPLIT_INV, acc[i][j][e]);
    }
    PROBE_STAMP(pr_end)
    if (p.splitk > 1) {
        if (!splitk_combine<TM, TN, BM * BN>(p, acc, tw, split, lds, t, wave, lane)) {
            unsigned int sink = 0;                   // this block is done: its slice of the next weights, then exit
            touch_next_weights(p, sink);
            retire_touches(sink);
            return;
        }
    }
    if constexpr (XA) {
        // Query projection of the text-context attention: the tile is 128 queries x the 64 columns of ONE head.  Leave it
        // in LDS as bf16 (exactly what the unfused path stores and reads back) and run the 16-row attention core on it:
        // the attention launch, its Q round trip through HBM and one kernel boundary disappear (70 per denoise step).
        static_assert(LNF && !GEGLU && !CONV && BM == 128 && BN == 64 && NW == 8 && sizeof(T) == 2, "xattn epilogue: 128 x 64 tile, 8 waves, 16-bit elements");
        static_assert(STAGES * STAGE >= 16384 + 3 * 16384, "xattn epilogue: Q tile + K/V ring fit the GEMM's ring");
        const float2* lnst = reinterpret_cast<const float2*>(lds + STAGES * STAGE);
        float mean[TM], rstd[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float2 v = lnst[wm * WTM + i * 16 + r16];
            mean[i] = v.x; rstd[i] = v.y;
        }
        __syncthreads();                               // every wave has read its last fragments: the ring is free
        T* qt = reinterpret_cast<T*>(lds);             // [128][64]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * WTN + j * 16 + 4 * q;
            const f32x4 cv = *reinterpret_cast<const f32x4*>(p.ln_c + n0 + col), dv = *reinterpret_cast<const f32x4*>(p.ln_d + n0 + col);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WTM + i * 16 + r16;
                typename V16<T>::x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (T)ln_fold(acc[i][j][e], mean[i], rstd[i], cv[e], dv[e]);
                *reinterpret_cast<typename V16<T>::x4*>(qt + row * 64 + col) = o4;
            }
        }
        __syncthreads();
        const int bimg = m0 / p.xa_T, head = tile_n;
        const T* Kb = (const T*)p.xa_k + (size_t)bimg * p.xa_S * p.xa_ldk + (size_t)head * 64;
        const T* Vb = (const T*)p.xa_v + (size_t)bimg * p.xa_S * p.xa_ldv + (size_t)head * 64;
        attn16_core<T, 8>(qt, 64, 128, Kb, Vb, p.xa_ldk, p.xa_ldv, p.xa_S, (T*)p.C + (size_t)m0 * p.ldc + (size_t)head * 64, p.ldc,
                       min(BM, p.M - m0), p.xa_scale_log2e, lds + 16384, wave, lane);
        unsigned int sink = 0;
        touch_next_weights(p, sink);
        retire_touches(sink);
    } else if constexpr (BM * BN >= 128 * 128 || GEGLU || CONV) {
        // staged epilogue (through LDS): the wide tiles, every GEGLU tile, the implicit-GEMM convs
        // (the LayerNorm (mean, rstd) rows sit behind the ring, which the staged tile takes over)
#ifdef ST_PROBE
        unsigned long long ept[6] = {0, 0, 0, 0, 0, 0};
#else
        unsigned long long* const ept = nullptr;
#endif
        staged_epilogue<TO, BM, BN, WGM, WGN, TM, TN, GEGLU, STAGES * STAGE, !LNF, is_fp8<T>()>(
            p, acc, m0, n0, tile_n, wm, r16, q, ColsPlain{wn, WTN}, lds, reinterpret_cast<const float2*>(lds + STAGES * STAGE), ept);
#ifdef ST_PROBE
        pr_x = ept[0] - pr_end; pr_d = ept[1] - ept[0];
#endif
    } else if constexpr (LNF) {
        // the small dense tiles (64 x 64 ... 128 x 80) keep the fragment-layout epilogue: with 2-4 accumulator tiles per wave
        // the two block barriers and the LDS round trip of the staged form cost more than its coalescing returns
        // (128 x 64: 4400 against 3700 cycles; from 128 x 128 on the staged form is level or ahead: tools/gemm_probe.py)
        float mean[TM], rstd[TM];
        const float2* lnst = reinterpret_cast<const float2*>(lds + STAGES * STAGE);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float2 v = lnst[wm * WTM + i * 16 + r16];
            mean[i] = v.x; rstd[i] = v.y;
        }
        // (specialised instances for the feature sets of the step, as in staged_epilogue)
        const bool inside = (m0 + BM <= p.M) && (n0 + BNO <= p.N) && (p.N & 3) == 0 && p.C != nullptr && !p.q8_out;
        const int flags = inside ? (((p.epi & ST_EPI_BIAS) ? EPI_F_BIAS : 0) | (p.epi & (ST_EPI_RESIDUAL | ST_EPI_ROWBIAS | ST_EPI_SILU) ? 1024 : 0) |
                                    (p.col_scale ? EPI_F_SCALE : 0) | EPI_F_LN) : -1;
        if (!is_fp8<T>() && flags == EPI_F_LN) gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, 0, 0, false, EPI_F_LN>(p, acc, m0, n0, wm, wn, r16, q, split, mean, rstd);
        else if (!is_fp8<T>() && flags == (EPI_F_LN | EPI_F_BIAS))
            gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, 0, 0, false, EPI_F_LN | EPI_F_BIAS>(p, acc, m0, n0, wm, wn, r16, q, split, mean, rstd);
        else if (is_fp8<T>() && flags == (EPI_F_LN | EPI_F_SCALE))
            gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, 0, 0, false, EPI_F_LN | EPI_F_SCALE>(p, acc, m0, n0, wm, wn, r16, q, split, mean, rstd);
        else gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU>(p, acc, m0, n0, wm, wn, r16, q, split, mean, rstd);
    } else {
#ifdef ST_PROBE
        unsigned long long ept[6] = {0, 0, 0, 0, 0, 0};
        gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, WGM, WGN>(p, acc, m0, n0, wm, wn, r16, q, split, nullptr, nullptr, lds, tile_n, ept);
        pr_x = ept[0] - pr_end; pr_d = ept[1] - ept[0];
#else
        const bool inside = (m0 + BM <= p.M) && (n0 + BNO <= p.N) && (p.N & 3) == 0 && p.C != nullptr;
        const int flags = inside ? (((p.epi & ST_EPI_BIAS) ? EPI_F_BIAS : 0) | ((p.epi & ST_EPI_RESIDUAL) ? EPI_F_RES : 0) | ((p.epi & ST_EPI_ROWBIAS) ? EPI_F_RB : 0) |
                                    ((p.epi & ST_EPI_SILU) ? EPI_F_SILU : 0) | (p.col_scale ? EPI_F_SCALE : 0) | (p.ln_c ? EPI_F_LN : 0) | (p.row_stats ? EPI_F_ROWS : 0) |
                                    ((p.col_stats && (p.N & 3) == 0) ? EPI_F_COLS : 0) | (p.q8_out ? EPI_F_Q8 : 0)) : -1;
#define ST_FRAG_CASE(M)                                                                                                                        \
    case (M):                                                                                                                                  \
        gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, WGM, WGN, false, (M)>(p, acc, m0, n0, wm, wn, r16, q, split, nullptr, nullptr, lds, tile_n); \
        break;
        if constexpr (!is_fp8<T>()) {
            switch (flags) {
                ST_FRAG_CASE(EPI_F_BIAS)
                ST_FRAG_CASE(EPI_F_BIAS | EPI_F_RES)
                ST_FRAG_CASE(EPI_F_BIAS | EPI_F_ROWS)
                ST_FRAG_CASE(EPI_F_BIAS | EPI_F_RES | EPI_F_ROWS)
                ST_FRAG_CASE(EPI_F_BIAS | EPI_F_RES | EPI_F_ROWS | EPI_F_Q8)
                ST_FRAG_CASE(EPI_F_BIAS | EPI_F_RES | EPI_F_COLS)
                default: gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, WGM, WGN>(p, acc, m0, n0, wm, wn, r16, q, split, nullptr, nullptr, lds, tile_n);
            }
        } else {
            switch (flags) {
                ST_FRAG_CASE(EPI_F_BIAS | EPI_F_SCALE | EPI_F_RES | EPI_F_ROWS)
                ST_FRAG_CASE(EPI_F_BIAS | EPI_F_SCALE | EPI_F_RES | EPI_F_ROWS | EPI_F_Q8)
                default: gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, WGM, WGN>(p, acc, m0, n0, wm, wn, r16, q, split, nullptr, nullptr, lds, tile_n);
            }
        }
#undef ST_FRAG_CASE
#endif
    }
#ifdef ST_PROBE
    {
        PROBE_STAMP(pr_fin)
        if (p.probe && lane == 0) {
            unsigned long long pr_rt1;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pr_rt1)::"memory");
            unsigned long long* o = p.probe + ((size_t)blockIdx.x * NW + wave) * 12;
            o[0] = pr_a; o[1] = pr_b; o[2] = pr_c; o[3] = pr_end - pr_start; o[4] = pr_fin - pr_end; o[5] = pr_x; o[6] = pr_d; o[7] = nk;
            o[8] = pr_start - pr_k0; o[9] = pr_rt0; o[10] = pr_rt1; o[11] = 0;      // prologue cycles; 100 MHz wall clock at entry / exit
        }
    }
#endif
}

// =============================================================================
// gemm8p: 256-row tiles for the large Linear problems (the projections at batch >= 2, the GEGLU projection at any
// batch).  A 128 x 128 tile needs (128+128)*128 B of LDS fill per 0.21 us of MFMA work - more than the ~130 GB/s one
// CU pulls from its L2 - a 256-row tile about half of that.  Two shapes of the same kernel:
//     256 x 256: 8 waves = 2 (rows) x 4 (columns), wave tile 128 x 64;
//     256 x 160: 8 waves = 4 x 2, wave tile 64 x 80 - SDXL's widths are 5 * 2^k: 1024 x 10240 (the GEGLU projection at
//                batch 1) is 160 tiles of 256 x 256 but 256 of 256 x 160, one per CU.
//   * A K tile is four phases, one quadrant of the wave tile each: (A0,B0) (A0,B1) (A1,B1) (A1,B0), Ah = the two halves of
//     the wave's rows, B0 / B1 = its first ceil(TN/2) / last floor(TN/2) accumulator columns.  A phase reads only the
//     fragments it is the first to use (A0+B0, B1, A1, nothing).
//   * The two halves of the block's waves run half a phase apart (waves 4-7 pass one extra barrier first): while one
//     half multiplies, the other reads fragments and issues DMAs, on the same SIMDs - a software ping-pong with two raw
//     barriers per phase and no wave ever doing both at once.
//   * LDS: two K tiles, each as four regions (A0, A1, B0, B1: the rows all eight waves read in the same phase), filled by
//     LDS-DMA one region per phase (two 1-KiB pieces per wave; a region with fewer than sixteen pieces fills up with dummy
//     pieces so that every wave counts the same vmcnt), swizzled on the source side as in gemm_dma_kernel.  A region is
//     refilled two phases after its last read and waited for (counted vmcnt, never 0) one phase before its first read,
//     which leaves four regions in flight at all times.
// Staged epilogue (GEGLU: tile columns [values | gates]), LayerNorm folding, statistics, next-weights touches and the
// XCD-aware tile order are the ones of gemm_dma_kernel.  No K split.
// =============================================================================
template <typename T, bool GEGLU, bool LNF, int BN = 256, int WGM = 2, int WGN = 4>
__global__ __launch_bounds__(512) void gemm8p_kernel(const GemmArgs p) {
    static_assert(sizeof(T) <= 2, "16-bit elements (bf16 / f16) or e4m3 bytes");
    constexpr int BM = 256, NW = 8;
    constexpr int KB = 128 / (int)sizeof(T);           // elements per 128-byte row of a K tile: 64, or 128 e4m3
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
    typedef typename std::conditional<sizeof(T) == 1, u32x4, Frag>::type Half;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const lnrows = lds + 2 * TILE_B;            // LayerNorm (mean, rstd) per row
    char* const dump = lnrows + BM * 8;               // target of the dummy DMAs

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef ST_PROBE
    unsigned long long pr_k0 = probe_now(), pr_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pr_rt0)::"memory");
#endif
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
    auto w_row = [&](int c) { return GEGLU ? (c < BN / 2 ? (size_t)(n0 + c) : (size_t)p.N + n0 + (c - BN / 2)) : (size_t)(n0 + c); };
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = (2 * wave + e) * 8 + lr;
        a_src[e] = Ap + (size_t)(m0 + (idx / (TMH * 16)) * WTM + (idx % (TMH * 16))) * p.lda + lc * EV;      // half h adds TMH*16 rows
        const int i0 = idx < RB0 ? idx : 0, i1 = idx < RB1 ? idx : 0;
        b0_src[e] = Wp + w_row((i0 / (TN0 * 16)) * WTN + (i0 % (TN0 * 16))) * p.K + lc * EV;
        b1_src[e] = Wp + w_row((i1 / (TN1 * 16)) * WTN + TN0 * 16 + (i1 % (TN1 * 16))) * p.K + lc * EV;
    }
    const size_t a_half = (size_t)(TMH * 16) * p.lda;
    const int nk = p.K / KB;

    // region r of a K tile: 0 = A0, 1 = A1, 2 = B0, 3 = B1.  Always two DMAs per wave: pieces beyond the region's rows and
    // `kt >= nk` are dummies (a zero line into the dump area).
    auto issue_half = [&](int kt, int region) {
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
            touch(p.bias, (long)n0 * 2, BNO * 2);
            if (GEGLU) touch(p.bias, ((long)p.N + n0) * 2, BNO * 2);
        }
        if (LNF) {
            touch(p.ln_c, (long)n0 * 4, BNO * 4); touch(p.ln_d, (long)n0 * 4, BNO * 4);
            if (GEGLU) { touch(p.ln_c, ((long)p.N + n0) * 4, BNO * 4); touch(p.ln_d, ((long)p.N + n0) * 4, BNO * 4); }
        }
        if (p.epi & ST_EPI_RESIDUAL) {
            constexpr int lines = (BNO * 2 + 127) / 128;
            for (int o = t; o < BM * lines; o += 512) {
                const int r = o / lines, l = o - r * lines;
                touch_at((const char*)p.residual + ((size_t)(m0 + r) * p.ldr + n0) * 2 + l * 128);
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
    // prologue: K tile 0 whole and A0, B0 of K tile 1 (the loop issues B1(1), A1(1), A0(2), B0(2), B1(2), ...)
    issue_half(0, 0); issue_half(0, 2); issue_half(0, 3); issue_half(0, 1); issue_half(1, 0); issue_half(1, 2);
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
    wait_vmcnt<8>();                                 // A0(0), B0(0) have landed (and every load older than the DMAs)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::"v"(touch_sink));
    if (wave >= 4) __builtin_amdgcn_s_barrier();     // the second half of the waves runs one barrier (half a phase) behind the first

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
    // A half in use, B0 (kept for the fourth phase), B1.  16-bit: [..][kk] = the operand of k step kk; e4m3: [..][0] is the whole
    // 128-k operand, assembled from the two 16-byte reads (chunks q and q + 4) where they land
    constexpr int NKK = sizeof(T) == 1 ? 1 : 2;
    Frag fa[TMH][NKK], fb0[TN0][NKK], fb1[TN1][NKK];
    auto read_op = [&](const char* base, const int (&off)[2], Frag (&dst)[NKK]) {
        if constexpr (sizeof(T) == 1) {
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
    auto read_b0 = [&](const char* tile) {
#pragma unroll
        for (int j = 0; j < TN0; ++j) read_op(tile + j * 2048, b0_off, fb0[j]);
    };
    auto read_b1 = [&](const char* tile) {
#pragma unroll
        for (int j = 0; j < TN1; ++j) read_op(tile + j * 2048, b1_off, fb1[j]);
    };
    auto quadrant = [&](auto mh_, auto nh_) {
        constexpr int mh = decltype(mh_)::value, nh = decltype(nh_)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
            for (int i = 0; i < TMH; ++i) {
                if constexpr (nh == 0) {
#pragma unroll
                    for (int j = 0; j < TN0; ++j) Mma<T>::run(acc[mh * TMH + i][j], fb0[j][kk], fa[i][kk]);
                } else {
#pragma unroll
                    for (int j = 0; j < TN1; ++j) Mma<T>::run(acc[mh * TMH + i][TN0 + j], fb1[j][kk], fa[i][kk]);
                }
            }
        __builtin_amdgcn_s_setprio(0);
    };
    // one phase: [fragment reads] [one region of DMA] [counted wait] barrier [MFMAs of one quadrant] barrier
    // after the issue of phase ph the eight DMAs of phases ph-3 .. ph may stay in flight: the region issued in
    // phase ph-4 has landed for this wave, and for everybody once both halves of the waves have passed their next barrier
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
    PROBE_STAMP(pr_start)
    for (int kt = 0; kt < nk; ++kt) {
        const char* tile = lds + (kt & 1) * TILE_B;
        // phase 0: (A0, B0); refill B1 of tile kt+1 (last read in phase 1 of tile kt-1)
        read_a(tile, 0); read_b0(tile);
        issue_half(kt + 1, 3);
        ST_PHASE_SYNC();
        quadrant(I0{}, I0{});
        ST_PHASE_END();
        // phase 1: (A0, B1); refill A1 of tile kt+1 (last read in phase 2 of tile kt-1)
        read_b1(tile);
        issue_half(kt + 1, 1);
        ST_PHASE_SYNC();
        quadrant(I0{}, I1{});
        ST_PHASE_END();
        // phase 2: (A1, B1); refill A0 of tile kt+2 (last read in phase 0 of this tile)
        read_a(tile, 1);
        issue_half(kt + 2, 0);
        ST_PHASE_SYNC();
        quadrant(I1{}, I1{});
        ST_PHASE_END();
        // phase 3: (A1, B0) from registers; refill B0 of tile kt+2 (last read in phase 0 of this tile)
        issue_half(kt + 2, 2);
        ST_PHASE_SYNC();
        quadrant(I1{}, I0{});
        ST_PHASE_END();
    }
#undef ST_PHASE_SYNC
#undef ST_PHASE_END
    if (wave < 4) __builtin_amdgcn_s_barrier();      // barrier counts of the two halves are equal again
    wait_vmcnt<0>();                                  // no LDS-DMA may outlive the workgroup's LDS allocation
    __builtin_amdgcn_s_barrier();
#ifdef ST_8P_NOEPI      // timing experiment: K loop only
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
#endif
#ifdef ST_PROBE
    PROBE_STAMP(pr_end)
    unsigned long long ept[6] = {0, 0, 0, 0, 0, 0};
    staged_epilogue<TO, BM, BN, WGM, WGN, TM, TN, GEGLU, 2 * TILE_B, !LNF, sizeof(T) == 1>(p, acc, m0, n0, tile_n, wm, r16, q, ColsPlain{wn, WTN}, lds,
                                                                        reinterpret_cast<const float2*>(lnrows), ept);
    {
        PROBE_STAMP(pr_fin)
        if (p.probe && lane == 0) {
            unsigned long long pr_rt1;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pr_rt1)::"memory");
            unsigned long long* o = p.probe + ((size_t)blockIdx.x * NW + wave) * 12;
            // epilogue: entry -> first barrier, -> chunk 0 parked + barrier, -> chunk 0 processed, -> its closing barrier, rest
            o[0] = ept[3] - ept[2]; o[1] = ept[0] - ept[3]; o[2] = ept[4] - ept[0]; o[3] = pr_end - pr_start; o[4] = pr_fin - pr_end;
            o[5] = ept[5] - ept[4]; o[6] = ept[1] - ept[5]; o[7] = nk; o[8] = pr_start - pr_k0; o[9] = pr_rt0; o[10] = pr_rt1; o[11] = 1;
        }
    }
#else
    staged_epilogue<TO, BM, BN, WGM, WGN, TM, TN, GEGLU, 2 * TILE_B, !LNF, sizeof(T) == 1>(p, acc, m0, n0, tile_n, wm, r16, q, ColsPlain{wn, WTN}, lds,
                                                                        reinterpret_cast<const float2*>(lnrows));
#endif
}

// the two shapes of gemm8p: 256 (2 x 4 waves) and 160 columns (4 x 2 waves)
static inline bool gemm8p_applies(const GemmArgs& a, int bn, int kb = 64) {
    const long n_rows = (a.epi & ST_EPI_GEGLU) ? 2L * a.N : a.N;
    return a.M % 256 == 0 && n_rows % bn == 0 && a.K % kb == 0 && a.K >= 2 * kb && a.N % 8 == 0 &&
           !(a.epi & ST_EPI_ROWBIAS) && (!a.row_stats || !(a.epi & ST_EPI_GEGLU));
}

template <typename T, bool GEGLU, bool LNF, int BN, int WGM, int WGN>
static void gemm8p_go(const GemmArgs& a, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)(2 * 128 * 128 + BN * 128) + 256 * 8 + 1024;
    auto kfn = gemm8p_kernel<T, GEGLU, LNF, BN, WGM, WGN>;
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

template <typename T, int BN, int WGM, int WGN>
static void gemm8p_launch(const GemmArgs& a, hipStream_t st) {
    const bool geglu = a.epi & ST_EPI_GEGLU;
    if (a.ln_c) { if (geglu) gemm8p_go<T, true, true, BN, WGM, WGN>(a, st); else gemm8p_go<T, false, true, BN, WGM, WGN>(a, st); }
    else { if (geglu) gemm8p_go<T, true, false, BN, WGM, WGN>(a, st); else gemm8p_go<T, false, false, BN, WGM, WGN>(a, st); }
}

#include "gemm4w.h"
// (instantiated in gemm_4w.hip only)
void gemm4w_bf16(const GemmArgs& a, hipStream_t st);
void gemm4w_f16(const GemmArgs& a, hipStream_t st);
void gemm4w_fp8(const GemmArgs& a, hipStream_t st);
template <typename T> static inline void gemm4w_call(const GemmArgs& a, hipStream_t st) {
    if constexpr (std::is_same<T, bf16>::value) gemm4w_bf16(a, st);
    else if constexpr (std::is_same<T, f16>::value) gemm4w_f16(a, st);
    else gemm4w_fp8(a, st);
}

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
    static_assert(sizeof(T) == 2, "16-bit elements (bf16 / f16)");
    constexpr int W = 1 << WL2, BM = TH * W, NW = WGM * WGN;
    static_assert(NW == 8 && BM % (16 * WGM) == 0 && BN % (16 * WGN) == 0 && W >= 16, "tile / wave layout");
    static_assert(!UPS || TH % 2 == 0, "upsampled tiles start on an even output row");
    constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 16, TN = WTN / 16, KB = 64;
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
        pa_ptr[e] = ok ? Xb + ((size_t)y * WI + x) * p.Cin + lc * 8 : nullptr;
    }
    const T* pb_ptr[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int row = (wave + j * NW) * 8 + lr;
        const int wrow = n0 + row;
        pb_ptr[j] = (wrow < p.N && wave + j * NW < B_PIECES) ? Wp + (size_t)wrow * p.K + ((lane & 7) ^ lr) * 8 : nullptr;
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
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
            // both 32-wide K halves are read up front: the second half's LDS latency hides under the first half's MFMAs
            Frag fa[2][TM], fb[2][TN];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int c = 4 * g + q;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    int pp;
                    if constexpr (UPS) {
                        const int uy = ty0 + (ppl[i] >> 16) + r - 1, ux = (ppl[i] & 0xffff) + s_ - 1;
                        pp = ((uy >> 1) - iy_base) * PWD + (ux >> 1) + 1;
                    } else {
                        pp = ppl[i] + r * PWD + s_;
                    }
                    fa[g][i] = *reinterpret_cast<const Frag*>(patch + pp * 128 + ((c ^ (pp & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[g][j] = *reinterpret_cast<const Frag*>(wt + rbl[j] * 128 + ((c ^ (rbl[j] & 7)) << 4));
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], fb[g][j], fa[g][i]);
            // pin the emitted order: DMAs, then every fragment read, then the MFMAs -
            // left alone hipcc sinks each read to just before its first use and waits lgkmcnt(0) a dozen times per trip
            __builtin_amdgcn_sched_group_barrier(0x020, n_p + B_IT, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);
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
    if (p.splitk > 1) {
        if (!splitk_combine<TM, TN, BM * BN>(p, acc, tw, split, lds, t, wave, lane)) {
            unsigned int sink = 0;                   // this block is done: its slice of the next weights, then exit
            touch_next_weights(p, sink);
            retire_touches(sink);
            return;
        }
    }
    staged_epilogue<T, BM, BN, WGM, WGN, TM, TN, false, 2 * PB + STAGES * WT_B, true>(p, acc, m0, n0, tile_n, wm, r16, q, ColsPlain{wn, WTN}, lds,
                                                                                      nullptr);
}

template <typename T, int BM, int BN, int WGM, int WGN, bool CONV>
static void launch_cfg(const GemmArgs& a, hipStream_t st) {
    const size_t lds = 2 * (size_t)(BM + BN) * 128;
    const int tiles_m = cdiv(a.M, BM);
    if constexpr (!CONV) {
        if (a.epi & ST_EPI_GEGLU) {
            hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WGM, WGN, CONV, true>), dim3(tiles_m * cdiv(a.N, BN / 2)),
                               dim3(WGM * WGN * 64), lds, st, a);
            return;
        }
    }
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WGM, WGN, CONV, false>), dim3(tiles_m * cdiv(a.N, BN)), dim3(WGM * WGN * 64), lds, st, a);
}

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

template <typename T, bool CONV>
static int gemm_dispatch(const GemmArgs& a, hipStream_t st) {
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
        for (const Cand& c : cands) {
            if (frag2<T>() && c.bn == 320) continue;                     // fp8 / split fragments are 32 bytes: the 320-wide wave tiles spill
            if (is_split<T>() && (c.cfg == CFG_256x128_W8 || c.cfg == CFG_128x160_W8)) continue;      // two accumulator sets: wave tiles of at most 8 x 16 x 16
            if ((a.epi & ST_EPI_GEGLU) && c.bn % 32 != 0) continue;      // GEGLU: the value / gate halves of a tile are whole accumulator tiles (BN = 80 is not)
            const long nt = tiles(c.bm, c.bn);
            const double trip = c.trip_us * (CONV ? 1.6 : 1.0);
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
                const double cost = rounds * (cdiv(nk, k_) * (trip > trip_bw ? trip : trip_bw) + 3.0) * (rounds > 1.0 ? 1.3 : 1.0) + (k_ > 1 ? 2.1 + 0.4 * slab_mb : 0.0);
                if (cost < best) { best = cost; cfg = c.cfg; sk = k_; }
            }
        }
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
            const bool take256 = f == CFG_256x256_8P || (f < 0 && c256 <= c160 && c256 < 1.1 * best);
            const bool take160 = f == CFG_256x160_8P || (f < 0 && c160 < c256 && c160 < 1.1 * best);
            if (take256 && c256 < 1e29) { gemm8p_launch<T, 256, 2, 4>(a, st); return st_check_launch(who); }
            if (take160 && c160 < 1e29) { gemm8p_launch<T, 160, 4, 2>(a, st); return st_check_launch(who); }
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
            case CFG_128x160_W8: launch_dma<T, 128, 160, 8, 1, 4, 1, CONV>(b, st); break;
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
static inline void take_hint(GemmArgs& a, const void* next_w, size_t next_bytes) {
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

// ---- host side of conv_halo_kernel --------------------------------------------------------------
static inline bool conv_halo_applies(const GemmArgs& a, int R, int ups) {
    static const bool off = dev_env_int("ST_CONV_HALO", 1) == 0;
    if (off) return false;
    if (R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1) return false;
    if (a.Cin % 64 != 0 || a.N % 4 != 0 || a.N < 64) return false;
    if (ups) {           // output 64 or 128 pixels wide, tiles of 256 output pixels
        if (a.Wout != 2 * a.Win || a.Hout != 2 * a.Hin || (a.Wout != 64 && a.Wout != 128)) return false;
        return a.Hout % (256 / a.Wout) == 0 && a.M % 256 == 0;
    }
    if (a.Win != 32 && a.Win != 64 && a.Win != 128) return false;
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
    const int bm = (!ups && a.Win == 128) ? 128 : 256, bn = (ups || a.Win == 128) ? 160 : 128;
    const int tiles = (a.M / bm) * cdiv(a.N, bn);
    const int ncs = a.Cin / 64;
    // K split over channel slices: aim at one round of ~240 blocks; a slice keeps at least two channel slices
    int sk = 1;
    if (a.partial && tiles < 200) {
        static const int target = dev_env_int("ST_HALO_BLOCKS", 240);
        sk = (target + tiles / 2) / tiles;
        if (sk > ncs / 2) sk = ncs / 2;
        if (sk < 1) sk = 1;
        while (sk > 1 && ((size_t)sk * tiles * bm * bn * 4 + 65536 > a.partial_bytes || tiles > 16384)) --sk;
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
    if (ups && a.Wout == 64) conv_halo_go<T, 6, 4, 160, 4, 2, true>(b, tiles * sk, st);
    else if (ups) conv_halo_go<T, 7, 2, 160, 4, 2, true>(b, tiles * sk, st);
    else if (a.Win == 32) conv_halo_go<T, 5, 8, 128, 4, 2>(b, tiles * sk, st);
    else if (a.Win == 64) conv_halo_go<T, 6, 4, 128, 4, 2>(b, tiles * sk, st);
    else conv_halo_go<T, 7, 1, 160, 4, 2>(b, tiles * sk, st);
    return st_check_launch("conv2d(halo)");
}

// ---- direct conv for thin inputs (conv_in: Cin = 4, K = R*S*Cin = 36) ---------------------
// Weights sit in LDS as fp32 [K][Cout]; a thread owns one output pixel and a strip of 16 output
// channels at a time: its K input values stay in registers, weight reads are wave-wide broadcasts
// (all lanes of a wave work on the same channel strip).
template <typename T, int KMAX>
__global__ __launch_bounds__(256) void conv_thin_kernel(const GemmArgs p, int R, int chunk) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];       // [K][chunk]: this block's output channels
    const T* x = (const T*)p.A;
    const T* w = (const T*)p.W;
    const int K = p.K, N = p.N;
    const int n_lo = blockIdx.y * chunk, n_hi = min(N, n_lo + chunk);  // blockIdx.y splits the output channels
    for (int i = threadIdx.x; i < K * (n_hi - n_lo); i += 256) {
        const int nl = i / K, k = i - nl * K;
        wsm[k * chunk + nl] = Elem<T>::to_f(w[(size_t)(n_lo + nl) * K + k]);
    }
    __syncthreads();
    const int m = blockIdx.x * 256 + threadIdx.x;
    const bool live = m < p.M;
    const int mm = live ? m : 0;
    const int hw = p.Hout * p.Wout;
    const int img = mm / hw, rem = mm - img * hw;
    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
    float xin[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) xin[k] = 0.f;
    if (R == 3 && p.S == 3 && p.Cin == 4 && KMAX >= 36) {
        // the SDXL conv_in shape, fully unrolled: static register indices, one 4-channel load per tap
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                int iy = oy * p.stride - p.pad + r, ix = ox * p.stride - p.pad + s_;
                const int He = p.ups ? 2 * p.Hin : p.Hin, We = p.ups ? 2 * p.Win : p.Win;
                const bool ok = iy >= 0 && ix >= 0 && iy < He && ix < We;
                if (p.ups) { iy >>= 1; ix >>= 1; }
                const T* xp = x + (((size_t)img * p.Hin + (ok ? iy : 0)) * p.Win + (ok ? ix : 0)) * 4;
                float f[4];
                Out4<T>::load(xp, f);
#pragma unroll
                for (int c = 0; c < 4; ++c) xin[(r * 3 + s_) * 4 + c] = ok ? f[c] : 0.f;
            }
    } else {
        int k = 0;
        for (int r = 0; r < R; ++r)
            for (int s_ = 0; s_ < p.S; ++s_) {
                int iy = oy * p.stride - p.pad + r, ix = ox * p.stride - p.pad + s_;
                const int He = p.ups ? 2 * p.Hin : p.Hin, We = p.ups ? 2 * p.Win : p.Win;
                const bool ok = iy >= 0 && ix >= 0 && iy < He && ix < We;
                if (p.ups) { iy >>= 1; ix >>= 1; }
                const T* xp = x + (((size_t)img * p.Hin + (ok ? iy : 0)) * p.Win + (ok ? ix : 0)) * p.Cin;
                for (int c = 0; c < p.Cin; ++c, ++k) {
                    const float v = ok ? Elem<T>::to_f(xp[c]) : 0.f;
#pragma unroll
                    for (int kk = 0; kk < KMAX; ++kk) if (kk == k) xin[kk] = v;     // keep xin[] in registers
                }
            }
    }
    for (int n0 = n_lo; n0 < n_hi; n0 += 16) {
        float acc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                const float xv = xin[k];
                const float* wr = wsm + k * chunk + (n0 - n_lo);
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + e);
                    acc[e] += xv * w4[0]; acc[e + 1] += xv * w4[1]; acc[e + 2] += xv * w4[2]; acc[e + 3] += xv * w4[3];
                }
            }
        }
        if (!live) continue;
#pragma unroll
        for (int e0 = 0; e0 < 16; e0 += 4) {
            const int co = n0 + e0;
            if (co >= N) break;
            float v[4] = {acc[e0], acc[e0 + 1], acc[e0 + 2], acc[e0 + 3]};
            if (p.epi & ST_EPI_BIAS)
                for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f(((const T*)p.bias)[co + e]);
            if (p.epi & ST_EPI_SILU)
                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
            if (p.epi & ST_EPI_ROWBIAS)
                for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f(((const T*)p.rowbias)[(size_t)(m / p.rows_per_batch) * N + co + e]);
            if (p.epi & ST_EPI_RESIDUAL)
                for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f(((const T*)p.residual)[(size_t)m * p.ldr + co + e]);
            Out4<T>::store((T*)p.C + (size_t)m * p.ldc + co, v);
        }
    }
}

template <typename T>
static int conv_thin_launch(const GemmArgs& a, int R, hipStream_t st) {
    // split the output channels over blockIdx.y until the launch has a few blocks per CU
    const int bx = cdiv(a.M, 256);
    int ny = cdiv(1024, bx);
    if (ny > a.N / 16) ny = a.N / 16;
    if (ny < 1) ny = 1;
    const int chunk = cdiv(cdiv(a.N, ny), 16) * 16;
    ny = cdiv(a.N, chunk);
    const size_t lds = (size_t)a.K * chunk * sizeof(float);
    ST_REQUIRE(lds <= 64 * 1024, "conv2d(thin): weights do not fit LDS");
    hipLaunchKernelGGL((conv_thin_kernel<T, 64>), dim3(bx, ny), dim3(256), lds, st, a, R, chunk);
    return st_check_launch("conv2d(thin)");
}

// ---- per-element-type runners: each is defined in exactly one translation unit (gemm_<what>_<type>.hip), so the kernels
// of one type compile beside those of the others; gemm_api.hip (the extern "C" entry points) only calls these.
int gemm_dense_bf16(const GemmArgs& a, hipStream_t st);
int gemm_dense_f16(const GemmArgs& a, hipStream_t st);
int gemm_dense_f32(const GemmArgs& a, hipStream_t st);
int gemm_dense_f32s(const GemmArgs& a, hipStream_t st);
int gemm_dense_fp8(const GemmArgs& a, hipStream_t st);
int gemm_xattn_bf16(const GemmArgs& a, hipStream_t st);
int gemm_xattn_f16(const GemmArgs& a, hipStream_t st);
int gemm_conv_bf16(const GemmArgs& a, int R, int ups, hipStream_t st);      // halo kernel when it applies, else implicit GEMM
int gemm_conv_f16(const GemmArgs& a, int R, int ups, hipStream_t st);
int gemm_conv_f32(const GemmArgs& a, hipStream_t st);
int gemm_conv_f32s(const GemmArgs& a, hipStream_t st);
int conv_thin_run(const GemmArgs& a, int R, int dtype, hipStream_t st);
