// GEMM-shaped operators, part 1 of the family headers (gemm_core.h includes them all): the launch descriptor (GemmArgs),
// the matrix-instruction traits per element type (Mma<bf16 / f16 / float / f8 / fsp>), the block -> tile map prepared on
// the host, the next-weights touches and a few device helpers shared by every kernel of the family.  Internal to csrc/.
#pragma once
#include "common.h"
#include "attention_core.h"
#include "split.h"
#include <stdlib.h>
#include <type_traits>

struct GemmArgs {
    const void* A; const void* W; const void* bias; const void* residual; const void* rowbias; void* C;
    int M, N, K;                 // N = output columns (with GEGLU: W has 2N rows)
    int Ng;                      // GEGLU: rows between a value row of W (bias, ln_c, ln_d, col_scale entry) and its gate row - N of the whole
                                 //   projection; a launch over a COLUMN RANGE of it (gemm_dispatch's column split) has N < Ng.  0 = N.
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
    unsigned next_row_lines;     // strided touch (a large matrix: only the leading K columns of every row): the row length in 128-byte lines,
    unsigned next_lead_shift;    //   2^next_lead_shift lines touched per row; next_bytes then counts the TOUCHED bytes (0: every line of next_bytes)
    // fp8 operands (st_linear_fp8): acc * row_scale[m] * col_scale[n] before anything else (col_scale has 2N entries with GEGLU)
    const float* row_scale; const float* col_scale;
    int rs_stride;               // stride of row_scale: 1 = a scale per row, 0 = one scale for the whole activation tensor
    // e4m3 copy of the output for an fp8 consumer (delayed per-tensor scaling, fp8.hip): q8[m][n] = e4m3(value * *q8_inv_scale),
    // the launch's max |value| goes to the q8_amax partial slots; C may then be NULL (only the copy is wanted)
    void* q8_out; long q8_ld; const float* q8_inv_scale; unsigned int* q8_amax;
    // split image of the output (strict mode, csrc/split.h): (M, N) values as fp16 pairs for a GEMM-shaped consumer, N % 32 == 0;
    // written beside C by the fp32 epilogues (st_arm_split_output)
    void* sp_out;
    const void* next_w; size_t next_bytes;   // weights of the NEXT launch (host hint): touched during this epilogue
    int helper_blocks;           // > 0: that many extra blocks at the end of the grid (idle CUs) do the touching instead
    // st_ln_linear_xattn: the tile is the query block of ONE head; its epilogue runs the text-context attention on it
    const void* xa_k; const void* xa_v; long xa_ldk, xa_ldv; int xa_S, xa_T; float xa_scale_log2e;
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


// The next launch's weights (the `next_weights` argument of the entry points) are touched one dword per 128-byte line, each block
// its slice, so that they sit in the memory-side cache when that launch starts (cold weights cost a GEMM 2-10 us:
// DESIGN.md section 6).  The loads are fire-and-forget: `sink` stays allocated until retire_touches(sink).
// touched line l -> line of the matrix: l itself, or (strided touch) line l mod 2^s of row l >> s
__device__ __forceinline__ size_t touch_line(const GemmArgs& p, size_t l) {
    if (!p.next_row_lines) return l;
    return (l >> p.next_lead_shift) * p.next_row_lines + (l & ((1u << p.next_lead_shift) - 1u));
}
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
        const char* a_ = (const char*)p.next_w + (touch_line(p, l) << 7);
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
int gemm_conv_f32s(const GemmArgs& a, int R, int ups, hipStream_t st);
int conv_thin_run(const GemmArgs& a, int R, int dtype, hipStream_t st);
