// GEMM-shaped operators: the LDS-DMA kernel (dense GEMM and implicit-GEMM conv; every element type), its row-statistics
// reader for folded LayerNorms and the in-launch split-K combine.  Internal to csrc/.
#pragma once
#include "epilogue.h"

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
            wrow = ncol + half * p.Ng;
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
            if (GEGLU) touch(p.bias, ((long)p.Ng + n0) * sizeof(TO), ncols_out * (int)sizeof(TO));
        }
        if (LNF) {
            touch(p.ln_c, (long)n0 * 4, ncols_out * 4); touch(p.ln_d, (long)n0 * 4, ncols_out * 4);
            if (GEGLU) { touch(p.ln_c, ((long)p.Ng + n0) * 4, ncols_out * 4); touch(p.ln_d, ((long)p.Ng + n0) * 4, ncols_out * 4); }
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
            } else {
                src = a_ptr[i] + (size_t)kt * a_adv[i];
            }
            const int pa = wave + i * NW;
            dma16_at<0>(src, (!UNEVEN || pa < A_PIECES) ? base + pa * 1024 : dump_addr);
        } else {
            const int j = i - A_IT;
            const int pb = wave + j * NW;
            const T* bsrc;
            if (CONV && U == 1) bsrc = b_ptr[j] + (b_adv[j] ? (size_t)((cs_r * p.S + cs_s) * p.Cin + cs_c0) : 0);     // W[n][tap][c]
            else bsrc = b_ptr[j] + (size_t)kt * b_adv[j];
            dma16_at<0>(bsrc, (!UNEVEN || pb < B_PIECES) ? base + A_BYTES + pb * 1024 : dump_addr);
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
        const char* sa = lds + buf * STAGE + (g / GPT) * TILE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[set][i] = read_frag(sa, wm * WTM + i * 16 + r16, g);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[set][j] = read_frag(sb, wn * WTN + j * 16 + r16, g);
    };
    auto mma_group = [&](int set) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (is_split<T>()) Mma<T>::run2(acc[i][j], corr[i][j], fb[set][j], fa[set][i]);
                else Mma<T>::run(acc[i][j], fb[set][j], fa[set][i]);
            }
    };

    int cur = 0, nxt = STAGES - 1;
    read_group(0, 0, 0);
    // The trip body is branch-free: trips past the last prefetch re-fetch the final stage into a
    // buffer nobody reads again, so the vmcnt bookkeeping is the same every trip.
    // PAR: which of the two fragment register sets group 0 of this trip multiplies from - with an odd number of groups
    // per stage (fp8: one) the sets trade places every trip, so the loop runs two trips per iteration.
    auto trip = [&](int kt, auto par_) {
        constexpr int PAR = decltype(par_)::value;
        const int pf = min(kt + STAGES - 1, nk - 1);      // stage to prefetch (clamped)
        auto group = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 1 < NG) {
                read_group(cur, g + 1, (g + 1 + PAR) & 1);
            } else {
                // stage kt+1 must have landed (own DMAs), then everyone's; the barrier also retires
                // every wave's reads of `cur` (all of them are in registers by now) before its refill
                // in flight at this point: stages kt+2 .. kt+S-2 whole, plus the shares of stage
                // kt+S-1 already issued by groups 0 .. NG-2 of this trip
                wait_vmcnt<EARLY ? 0 : (STAGES - 3) * G + dma_before_last_group(G, NG)>();
                __builtin_amdgcn_s_barrier();
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
            __builtin_amdgcn_sched_group_barrier(0x100, (TM + TN) * RPF, 0);
            if constexpr (n_dma > 0) __builtin_amdgcn_sched_group_barrier(0x020, n_dma, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN * mfma_per_frag<T>(), 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        group(std::integral_constant<int, 0>{});
        if constexpr (NG > 1) group(std::integral_constant<int, 1>{});
        if constexpr (NG > 2) {
            group(std::integral_constant<int, 2>{});
            group(std::integral_constant<int, 3>{});
        }
        // the prefetched first fragments of the next stage have had a whole MFMA group to land: retire
        // them here so the compiler enters the next trip with an empty LDS scoreboard (exact waits)
        __builtin_amdgcn_s_waitcnt(0xc07f);
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
        staged_epilogue<TO, BM, BN, WGM, WGN, TM, TN, GEGLU, STAGES * STAGE, !LNF, is_fp8<T>()>(
            p, acc, m0, n0, tile_n, wm, r16, q, ColsPlain{wn, WTN}, lds, reinterpret_cast<const float2*>(lds + STAGES * STAGE));
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
        const bool inside = (m0 + BM <= p.M) && (n0 + BNO <= p.N) && (p.N & 3) == 0 && p.C != nullptr && !p.q8_out && !p.sp_out;
        const int flags = inside ? (((p.epi & ST_EPI_BIAS) ? EPI_F_BIAS : 0) | (p.epi & (ST_EPI_RESIDUAL | ST_EPI_ROWBIAS | ST_EPI_SILU) ? 1024 : 0) |
                                    (p.col_scale ? EPI_F_SCALE : 0) | EPI_F_LN) : -1;
        if (!is_fp8<T>() && flags == EPI_F_LN) gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, 0, 0, false, EPI_F_LN>(p, acc, m0, n0, wm, wn, r16, q, split, mean, rstd);
        else if (!is_fp8<T>() && flags == (EPI_F_LN | EPI_F_BIAS))
            gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, 0, 0, false, EPI_F_LN | EPI_F_BIAS>(p, acc, m0, n0, wm, wn, r16, q, split, mean, rstd);
        else if (is_fp8<T>() && flags == (EPI_F_LN | EPI_F_SCALE))
            gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, 0, 0, false, EPI_F_LN | EPI_F_SCALE>(p, acc, m0, n0, wm, wn, r16, q, split, mean, rstd);
        else gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU>(p, acc, m0, n0, wm, wn, r16, q, split, mean, rstd);
    } else {
        const bool inside = (m0 + BM <= p.M) && (n0 + BNO <= p.N) && (p.N & 3) == 0 && p.C != nullptr;
        const int flags = inside ? (((p.epi & ST_EPI_BIAS) ? EPI_F_BIAS : 0) | ((p.epi & ST_EPI_RESIDUAL) ? EPI_F_RES : 0) | ((p.epi & ST_EPI_ROWBIAS) ? EPI_F_RB : 0) |
                                    ((p.epi & ST_EPI_SILU) ? EPI_F_SILU : 0) | (p.col_scale ? EPI_F_SCALE : 0) | (p.ln_c ? EPI_F_LN : 0) | (p.row_stats ? EPI_F_ROWS : 0) |
                                    ((p.col_stats && (p.N & 3) == 0) ? EPI_F_COLS : 0) | (p.q8_out ? EPI_F_Q8 : 0) | (p.sp_out ? EPI_F_SP : 0)) : -1;
#define ST_FRAG_CASE(M)                                                                                                                        \
    case (M):                                                                                                                                  \
        gemm_epilogue<TO, TM, TN, WTM, WTN, GEGLU, WGM, WGN, false, (M)>(p, acc, m0, n0, wm, wn, r16, q, split, nullptr, nullptr, lds, tile_n); \
        break;
        bool done = false;
        if constexpr (std::is_same<TO, float>::value) {      // (strict mode producers with the split image; anything else below)
            done = true;
            switch (flags) {
                ST_FRAG_CASE(EPI_F_BIAS | EPI_F_ROWS | EPI_F_SP)
                ST_FRAG_CASE(EPI_F_BIAS | EPI_F_RES | EPI_F_ROWS | EPI_F_SP)
                default: done = false;
            }
        }
        if (done) {
        } else if constexpr (!is_fp8<T>()) {
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
    }
}
