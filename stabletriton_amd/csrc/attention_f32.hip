// Attention of the strict (fp32-parity) mode on the 16-bit matrix pipe: out = softmax(q k^T * scale) v per head, fp32 in and
// out, head_dim 64.  Row A of SURVEY.md 8a in the precision of the reference's eager path (unet_pt.py:133-142: fp32 matmul,
// fp32 softmax, fp32 matmul).
//
// Every matrix operand is a split image (csrc/split.h: x ~ hi + lo * 2^-11, two IEEE halves, 22 significant bits) and every
// product three v_mfma_f32_16x16x32_f16 - hi.hi into a main accumulator, hi.lo and lo.hi into a correction accumulator that
// counts in units of 2^-11 - with fp32 accumulation; the softmax itself is fp32 (v_exp_f32).  Against the one-query-row-per-
// thread FMA kernel this replaces (57 ms of a 139-ms strict step): the 4096-token level 2.36 ms -> see DESIGN.md section 6.
//
// Structure (the 16-row flash kernel of attention_core.h, restated for split operands):
//   * a wave owns 16 query rows; Q is scaled by scale * log2(e) in fp32, THEN split, and stays in registers;
//   * scores transposed, S^T = K Q^T: a lane holds the scores of ONE query row (keys 16 kb + 4 g + r), so the row maximum
//     is in-lane plus one two-step lane exchange, and P^T is the B operand of O^T = V^T P^T straight from registers: the
//     probabilities are split in registers (two conversions and a fused multiply-add each);
//   * V^T fragments by ds_read_b64_tr_b16 from the row-major hi and lo images of the V tile;
//   * the row sums come out of the matrix pipe (a fifth "d block" of V^T that is 1 in its first row), in the same split
//     arithmetic as the numerator: what divides O is the sum of exactly the P values that multiplied V;
//   * lazy reference maximum (moves only when a tile outruns it by 2^6), S accumulators start at -m_ref;
//   * K / V tiles of 64 keys are read as fp32, split by the block's threads and stored into a two-buffer LDS ring (the loads
//     of tile t+1 fly under the matrix work of tile t): 64 KiB of LDS, two blocks per CU, one barrier per tile.
#include "attention_core.h"
#include "split.h"

namespace {

constexpr int SP_PLANE = ATT_KV * 128;            // one half-precision image of a 64-key tile: 64 rows of 64 halves
constexpr int SP_BUF = 4 * SP_PLANE;              // K hi, K lo, V hi, V lo

// PRE: K and V arrive as split images already (the q|k|v projection left them, st_arm_split_output): K / V point at the image
// rows' first byte of head 0's columns, ldk / ldv are the images' row lengths in VALUES (4 bytes each); the tile staging
// then moves halves (two 16-byte loads, two LDS stores per piece) and converts nothing.
template <int NW, bool PRE>
__global__ __launch_bounds__(NW * 64) void attn_split_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                             const float* __restrict__ V, float* __restrict__ O, int T, int S,
                                                             long ldq, long ldk, long ldv, long ldo, float scale_log2e, char* __restrict__ Os, int Cs, int H) {
    // Os != nullptr: also the split image of the output, rows = (batch, token), Cs = H * 64 values per row (the consumer is the
    // output projection: st_arm_split_output)
    constexpr int NT = NW * 64;
    constexpr int TASKS = 2 * ATT_KV * 8 / NT;     // (row, 8-value chunk) pieces of a K + V tile per thread
    static_assert(2 * ATT_KV * 8 % NT == 0 && TASKS % 2 == 0, "the K and the V pieces divide over the threads");
    extern __shared__ __attribute__((aligned(16))) char lds[];      // two buffers of SP_BUF bytes

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // 1-D grid, XCD-aware (as attn32i_kernel): consecutive block ids go round-robin to the eight XCDs, so XCD c takes the c-th
    // contiguous eighth of the (batch, head, query block) list and the query blocks of one head - they all stream that head's K
    // and V, 2 MB of split halves at 4096 keys - meet in ONE L2.  In (x, y, z) grid order every XCD streamed every head: the
    // 4096-token launch re-read 1.3 GB through the fabric, 5.4 TB/s, and was bound by exactly that.
    int head, b, xblk;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7;
        const int w = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
        const int gx = (T + 16 * NW - 1) / (16 * NW);
        const int hb = w / gx;
        xblk = w - hb * gx;
        b = hb / H;
        head = hb - b * H;
    }
    const int c16 = lane & 15, g = lane >> 4;
    const int q0 = xblk * NW * 16 + wave * 16;
    const int qrow = min(q0 + c16, T - 1);
    const float* Kb = K + (size_t)b * S * ldk + (size_t)head * ATT_D;      // (PRE: 64 values of an image row = 256 bytes = [hi32 | lo32 | hi32 | lo32])
    const float* Vb = V + (size_t)b * S * ldv + (size_t)head * ATT_D;

    // ---- Q fragments: d = 32 ks + 8 g .. + 7 of this lane's query row ------------------------------------------------
    f16x8 qh[2], ql[2];
    {
        const float* qp = Q + (size_t)b * T * ldq + (size_t)qrow * ldq + (size_t)head * ATT_D;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 32 * ks + 8 * g), c = *reinterpret_cast<const f32x4*>(qp + 32 * ks + 8 * g + 4);
            const float v[8] = {a[0] * scale_log2e, a[1] * scale_log2e, a[2] * scale_log2e, a[3] * scale_log2e,
                                c[0] * scale_log2e, c[1] * scale_log2e, c[2] * scale_log2e, c[3] * scale_log2e};
            split8(v, qh[ks], ql[ks]);
        }
    }
    f16x8 ones;                                    // V^T "row 64": one for the lanes that hold d = 0 of the extra block
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (f16)(c16 == 0 ? 1.0f : 0.0f);

    // ---- tile staging: piece i of this thread = (K or V, key row, chunk of 8 values); fp32 -> registers -> split -> LDS
    const int nkt = (S + ATT_KV - 1) / ATT_KV;
    f32x4 stg[TASKS][2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < TASKS; ++i) {
            const int id = t + (i % (TASKS / 2)) * NT;
            const int row = id >> 3, c = id & 7;
            const int key = kt * ATT_KV + row;
            const bool isv = i >= TASKS / 2;
            const float* rowp = isv ? Vb + (size_t)min(key, S - 1) * ldv : Kb + (size_t)min(key, S - 1) * ldk;
            if constexpr (PRE) {          // chunk c = values 8c .. 8c+7: hi halves at byte (c / 4) * 128 + (c % 4) * 16 of the row's 256, lo halves 64 further
                const char* src = reinterpret_cast<const char*>(rowp) + (c >> 2) * 128 + (c & 3) * 16;
                stg[i][0] = *reinterpret_cast<const f32x4*>(src);
                stg[i][1] = *reinterpret_cast<const f32x4*>(src + 64);
            } else {
                stg[i][0] = *reinterpret_cast<const f32x4*>(rowp + c * 8);
                stg[i][1] = *reinterpret_cast<const f32x4*>(rowp + c * 8 + 4);
            }
            if (key >= S) { stg[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; stg[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }      // masked keys: finite zeros
        }
    };
    auto store_tile = [&](int buf) {
        char* base = lds + buf * SP_BUF;
#pragma unroll
        for (int i = 0; i < TASKS; ++i) {
            const int id = t + (i % (TASKS / 2)) * NT;
            const int row = id >> 3, c = id & 7;
            const bool isv = i >= TASKS / 2;
            f16x8 hi, lo;
            if constexpr (PRE) {
                hi = __builtin_bit_cast(f16x8, stg[i][0]); lo = __builtin_bit_cast(f16x8, stg[i][1]);
            } else {
                const float v[8] = {stg[i][0][0], stg[i][0][1], stg[i][0][2], stg[i][0][3], stg[i][1][0], stg[i][1][1], stg[i][1][2], stg[i][1][3]};
                split8(v, hi, lo);
            }
            char* dst = base + (isv ? 2 * SP_PLANE : 0) + row * 128 + ((c ^ (isv ? swz_v16(row) : swz_k16(row))) << 4);
            *reinterpret_cast<f16x8*>(dst) = hi;
            *reinterpret_cast<f16x8*>(dst + SP_PLANE) = lo;
        }
    };

    // fragment offsets inside an image (the maps of attn16_core)
    const int k_off0 = c16 * 128 + (((0 + g) ^ swz_k16(c16)) << 4);
    const int k_off1 = c16 * 128 + (((4 + g) ^ swz_k16(c16)) << 4);
    const int vkey = 4 * g + (c16 >> 2);
    const int vsw = swz_v16(vkey);
    const int vrow = vkey * 128 + 8 * (c16 & 1);
    const int vbit = (c16 & 3) >> 1;
    const int v_off[4] = {vrow + (((0 ^ vsw) + vbit) << 4), vrow + (((2 ^ vsw) + vbit) << 4), vrow + (((4 ^ vsw) + vbit) << 4),
                          vrow + (((6 ^ vsw) + vbit) << 4)};

    f32x4 om[5], oc[5];                            // O^T d blocks 0..3 and the row-sum block: main / correction accumulators
#pragma unroll
    for (int i = 0; i < 5; ++i) { om[i] = f32x4{0.f, 0.f, 0.f, 0.f}; oc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float m_ref = 0.f;                             // reference maximum of this lane's query row (base-2 exponent units)

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const char* kb_ = lds + cur * SP_BUF;
        const char* vb_ = kb_ + 2 * SP_PLANE;
        if (kt + 1 < nkt) load_tile(kt + 1);       // in flight under this tile's matrix work

        // ---- fragment reads first, matrix work behind them: left to itself hipcc reads four K fragments, waits for them, multiplies,
        //      reads the next four ... and every group of six MFMAs starts with a full LDS round trip (a third of the trip).  All
        //      sixteen K fragments and the V^T fragments of the first key half go out here, the second half's behind the scores.
        f16x8 kh[4][2], kl[4][2], vh[4], vl[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            kh[kb][0] = *reinterpret_cast<const f16x8*>(kb_ + kb * 2048 + k_off0); kl[kb][0] = *reinterpret_cast<const f16x8*>(kb_ + SP_PLANE + kb * 2048 + k_off0);
            kh[kb][1] = *reinterpret_cast<const f16x8*>(kb_ + kb * 2048 + k_off1); kl[kb][1] = *reinterpret_cast<const f16x8*>(kb_ + SP_PLANE + kb * 2048 + k_off1);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            vh[db] = v_frag<f16>(vb_, v_off[db], v_off[db] + 2048);
            vl[db] = v_frag<f16>(vb_ + SP_PLANE, v_off[db], v_off[db] + 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- scores of this tile: s[kb][r] = key 16 kb + 4 g + r against this lane's query row, minus m_ref
        f32x4 s[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 mn = {-m_ref, -m_ref, -m_ref, -m_ref}, cr = {0.f, 0.f, 0.f, 0.f};
            mn = AttMma<f16>::m16(kh[kb][0], qh[0], mn);
            cr = AttMma<f16>::m16(kh[kb][0], ql[0], cr);
            cr = AttMma<f16>::m16(kl[kb][0], qh[0], cr);
            mn = AttMma<f16>::m16(kh[kb][1], qh[1], mn);
            cr = AttMma<f16>::m16(kh[kb][1], ql[1], cr);
            cr = AttMma<f16>::m16(kl[kb][1], qh[1], cr);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][r] = __builtin_fmaf(cr[r], 1.0f / ST_SPLIT_SCALE, mn[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
        f16x8 vh1[4], vl1[4];                       // V^T of the second key half: in flight under the softmax
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            vh1[db] = v_frag<f16>(vb_, v_off[db] + 4096, v_off[db] + 4096 + 2048);
            vl1[db] = v_frag<f16>(vb_ + SP_PLANE, v_off[db] + 4096, v_off[db] + 4096 + 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
        if ((kt + 1) * ATT_KV > S) {                // mask the tail keys (only the last tile has any)
            const int kbase = kt * ATT_KV + 4 * g;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kbase + 16 * kb + r >= S) s[kb][r] = -INFINITY;
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][r]);
        if (kt == 0 || __any(mx > ATT_LAG)) {
            // exact path: the row maximum (over the four lanes that share the row) becomes the reference of every row that
            // is on its first tile or has outrun the lag; O and the row sum, expressed against the old reference, follow
            const float rmx = xmax32(xmax16(mx));
            const float delta = ((kt == 0 || rmx > ATT_LAG) && rmx > -INFINITY) ? rmx : 0.f;
            const float alpha = kt == 0 ? 1.f : fast_exp2(-delta);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kb][r] -= delta;
#pragma unroll
            for (int db = 0; db < 5; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) { om[db][r] *= alpha; oc[db][r] *= alpha; }
            m_ref += delta;
        }
        // ---- P = 2^s, split in registers; O^T += V^T P^T
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = fast_exp2(s[2 * kp + (j >> 2)][j & 3]);
            f16x8 ph, pl;
            split8(pv, ph, pl);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const f16x8 a = kp ? vh1[db] : vh[db], c = kp ? vl1[db] : vl[db];
                om[db] = AttMma<f16>::m16(a, ph, om[db]);
                oc[db] = AttMma<f16>::m16(a, pl, oc[db]);
                oc[db] = AttMma<f16>::m16(c, ph, oc[db]);
            }
            om[4] = AttMma<f16>::m16(ones, ph, om[4]);
            oc[4] = AttMma<f16>::m16(ones, pl, oc[4]);
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);      // (buffer cur ^ 1 was last read in trip kt - 1: every wave is past that trip's barrier)
        __syncthreads();
    }

    // row sum: row 0 of the extra block lives in register 0 of the lanes with g == 0
    const float l = __shfl(__builtin_fmaf(oc[4][0], 1.0f / ST_SPLIT_SCALE, om[4][0]), c16, 64);
    const float inv = 1.0f / l;
    if (q0 + c16 < T) {
        float* orow = O + (size_t)b * T * ldo + (size_t)(q0 + c16) * ldo + (size_t)head * ATT_D;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            f32x4 a_;
#pragma unroll
            for (int e = 0; e < 4; ++e) a_[e] = __builtin_fmaf(oc[db][e], 1.0f / ST_SPLIT_SCALE, om[db][e]) * inv;
            *reinterpret_cast<f32x4*>(orow + 16 * db + 4 * g) = a_;
            if (Os) {
                const float v4[4] = {a_[0], a_[1], a_[2], a_[3]};
                split_store4(Os + ((size_t)b * T + q0 + c16) * Cs * 4, head * ATT_D + 16 * db + 4 * g, v4);
            }
        }
    }
}

}      // namespace

// (entry: st_attention with dtype ST_F32, attention.hip)
int attention_f32_launch(const float* q, const float* k, const float* v, float* out, int B, int T, int S, int H,
                         long ldq, long ldk, long ldv, long ldo, float scale, void* out_split, bool presplit, hipStream_t st) {
    const float c = scale * 1.4426950408889634f;
    constexpr size_t LDS = 2 * SP_BUF;
    // 64 query rows per block: the 1024-token level at batch 1 is 320 blocks, two per CU (LDS 64 KiB each)
    auto kfn = presplit ? attn_split_kernel<4, true> : attn_split_kernel<4, false>;
    ST_REQUIRE((long)cdiv(T, 64) * H * B < (1L << 31), "attention: too many blocks");
    hipLaunchKernelGGL(kfn, dim3(cdiv(T, 64) * H * B), dim3(256), LDS, st, q, k, v, out, T, S, ldq, ldk, ldv, ldo, c, (char*)out_split, H * ATT_D, H);
    return st_check_launch("attention");
}
