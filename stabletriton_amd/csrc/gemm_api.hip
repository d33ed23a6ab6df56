// extern "C" entry points of the GEMM-shaped operators (st_linear, st_ln_linear, st_ln_linear_xattn, st_linear_fp8,
// st_conv2d): argument checks and GemmArgs plumbing; the kernels live in gemm_core.h and are instantiated per element
// type in gemm_dense_*.hip / gemm_conv_*.hip / gemm_f32.hip / gemm_fp8.hip.
#include "gemm_core.h"

#ifdef ST_DEV_CONFIGS
int g_dbg_cfg = -1, g_dbg_fusek = -1;
extern "C" void st_debug_force_gemm(int cfg, int fusek) { g_dbg_cfg = cfg; g_dbg_fusek = fusek; }
#endif

static int run_dense(const GemmArgs& a, int dtype, hipStream_t st) {
    switch (dtype) {
        case ST_BF16: return gemm_dense_bf16(a, st);
        case ST_F16: return gemm_dense_f16(a, st);
        case ST_F32S: return gemm_dense_f32s(a, st);
        default: return gemm_dense_f32(a, st);
    }
}

static int check_q8(const char* who, GemmArgs& a, void* q8, long ldq8, const float* q8_inv_scale, unsigned int* q8_amax) {
    if (!q8) return 0;
    ST_REQUIRE(q8_inv_scale && q8_amax, "%s: the e4m3 copy needs its scale and its amax slots", who);
    ST_REQUIRE(a.N % 8 == 0 && ldq8 % 8 == 0 && (uintptr_t)q8 % 8 == 0, "%s: e4m3 copy: N and its row stride must be multiples of 8", who);
    a.q8_out = q8; a.q8_ld = ldq8; a.q8_inv_scale = q8_inv_scale; a.q8_amax = q8_amax;
    return 0;
}

static int linear_impl(const void* x, const void* W, const void* bias, const void* residual, const void* rowbias, void* y,
                       int M, int N, int K, long lda, long ldc, long ldr, int rows_per_batch, int epilogue, int dtype,
                       void* workspace, size_t workspace_bytes, float* row_stats, int row_stats_capacity,
                       int* row_stats_chunks, float* col_stats, int col_stats_tiles, int* col_stats_rows,
                       void* q8, long ldq8, const float* q8_inv_scale, unsigned int* q8_amax,
                       const void* next_weights, size_t next_weights_bytes, void* stream) {
    if (col_stats_rows) *col_stats_rows = 0;
    ST_REQUIRE(x && W && y, "linear: null pointer");
    ST_REQUIRE(M > 0 && N > 0 && K > 0, "linear: bad shape M=%d N=%d K=%d", M, N, K);
    ST_REQUIRE(st_dtype_ok_gemm(dtype), "linear: unsupported dtype %d", dtype);
    const int vec = st_dtype_is16(dtype) ? 8 : (dtype == ST_F32S ? 32 : 4);      // (split operands: whole 32-value segments)
    ST_REQUIRE(K % vec == 0 && lda % vec == 0, "linear: K=%d and lda=%ld must be multiples of %d", K, lda, vec);
    ST_REQUIRE(ldc % 4 == 0 && (!(epilogue & ST_EPI_RESIDUAL) || ldr % 4 == 0), "linear: ldc/ldr must be multiples of 4");
    ST_REQUIRE(((uintptr_t)x | (uintptr_t)W) % 16 == 0 && (uintptr_t)y % 16 == 0, "linear: pointers must be 16-byte aligned");
    GemmArgs a = {};
    a.A = x; a.W = W; a.bias = bias; a.residual = residual; a.rowbias = rowbias; a.C = y;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldr = ldr; a.rows_per_batch = rows_per_batch; a.epi = epilogue;
    a.splitk = 1; a.partial = (float*)workspace; a.partial_bytes = workspace ? workspace_bytes : 0;
    ST_REQUIRE(!row_stats || !(epilogue & ST_EPI_GEGLU), "linear: row_stats with GEGLU is not supported");
    a.row_stats = row_stats; a.stats_capacity = row_stats_capacity; a.stats_chunks_out = row_stats_chunks;
    a.col_stats = col_stats; a.col_tiles_cap = col_stats_tiles; a.col_rows_out = col_stats_rows;
    take_hint(a, next_weights, next_weights_bytes);
#ifdef ST_PROBE8
    { const char* e_ = getenv("ST_PROBE_KNOB"); a.korder = e_ ? atoi(e_) : 0; }      // (developer probe build only: gemm8p.h)
#endif
    if (int e = check_epilogue("linear", a)) return e;
    if (q8) {
        ST_REQUIRE(st_dtype_is16(dtype), "linear: the e4m3 copy is emitted by the 16-bit kernels");
        if (int e = check_q8("linear", a, q8, ldq8, q8_inv_scale, q8_amax)) return e;
    }
    if (int e = st_take_split_arm("linear", M, N, !st_dtype_is16(dtype) && N % 32 == 0, &a.sp_out)) return e;
    return run_dense(a, dtype, (hipStream_t)stream);
}

extern "C" int st_linear(const void* x, const void* W, const void* bias, const void* residual, const void* rowbias, void* y,
                         int M, int N, int K, long lda, long ldc, long ldr, int rows_per_batch, int epilogue, int dtype,
                         void* workspace, size_t workspace_bytes, float* row_stats, int row_stats_capacity,
                         int* row_stats_chunks, float* col_stats, int col_stats_tiles, int* col_stats_rows,
                         const void* next_weights, size_t next_weights_bytes, void* stream) {
    return linear_impl(x, W, bias, residual, rowbias, y, M, N, K, lda, ldc, ldr, rows_per_batch, epilogue, dtype, workspace, workspace_bytes,
                       row_stats, row_stats_capacity, row_stats_chunks, col_stats, col_stats_tiles, col_stats_rows, nullptr, 0, nullptr, nullptr,
                       next_weights, next_weights_bytes, stream);
}

// st_linear that also leaves an e4m3 copy of its output for an fp8 consumer (see the header).
extern "C" int st_linear_emit8(const void* x, const void* W, const void* bias, const void* residual, const void* rowbias, void* y,
                               int M, int N, int K, long lda, long ldc, long ldr, int rows_per_batch, int epilogue, int dtype,
                               void* workspace, size_t workspace_bytes, float* row_stats, int row_stats_capacity,
                               int* row_stats_chunks, float* col_stats, int col_stats_tiles, int* col_stats_rows,
                               void* q8, long ldq8, const float* q8_inv_scale, unsigned int* q8_amax,
                               const void* next_weights, size_t next_weights_bytes, void* stream) {
    ST_REQUIRE(q8, "linear_emit8: null e4m3 output");
    return linear_impl(x, W, bias, residual, rowbias, y, M, N, K, lda, ldc, ldr, rows_per_batch, epilogue, dtype, workspace, workspace_bytes,
                       row_stats, row_stats_capacity, row_stats_chunks, col_stats, col_stats_tiles, col_stats_rows, q8, ldq8, q8_inv_scale, q8_amax,
                       next_weights, next_weights_bytes, stream);
}

// LayerNorm folded into the following Linear (or GEGLU projection): see GemmArgs::ln_c.
extern "C" int st_ln_linear(const void* x, const float* row_stats, int row_stats_chunks, const void* Wg, const float* c,
                            const float* d, void* y, int M, int N, int K, long lda, long ldc, float eps, int epilogue,
                            int dtype, const void* next_weights, size_t next_weights_bytes, void* stream) {
    ST_REQUIRE(x && Wg && c && d && y && row_stats, "ln_linear: null pointer");
    ST_REQUIRE(row_stats_chunks > 0, "ln_linear: the producer emitted no row statistics");
    ST_REQUIRE(M > 0 && N > 0 && K > 0, "ln_linear: bad shape M=%d N=%d K=%d", M, N, K);
    ST_REQUIRE(st_dtype_ok_gemm(dtype), "ln_linear: unsupported dtype %d", dtype);
    const int kb = st_dtype_is16(dtype) ? 64 : 32;
    ST_REQUIRE(K % kb == 0 && lda % (dtype == ST_F32S ? 32 : kb / 8) == 0, "ln_linear: K=%d must be a multiple of %d", K, kb);
    ST_REQUIRE(ldc % 4 == 0, "ln_linear: ldc must be a multiple of 4");
    ST_REQUIRE((epilogue & ~ST_EPI_GEGLU) == 0, "ln_linear: only the GEGLU epilogue flag is accepted (bias lives in d)");
    ST_REQUIRE(((uintptr_t)x | (uintptr_t)Wg | (uintptr_t)y) % 16 == 0, "ln_linear: pointers must be 16-byte aligned");
    GemmArgs a = {};
    a.A = x; a.W = Wg; a.C = y; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.epi = epilogue;
    a.ln_c = c; a.ln_d = d; a.ln_eps = eps; a.splitk = 1; a.ln_stats = row_stats; a.ln_chunks = row_stats_chunks;
    take_hint(a, next_weights, next_weights_bytes);
    if (int e = st_take_split_arm("ln_linear", M, N, !st_dtype_is16(dtype) && N % 32 == 0, &a.sp_out)) return e;
    return run_dense(a, dtype, (hipStream_t)stream);
}

// The query projection of the text-context attention and that attention as ONE launch (transformer block:
// norm2 -> attn2.to_q -> attention over the 77 hoisted context keys, unet_pt.py:133-142,192-208):
//   out = softmax(LN(x) Wq^T (+bias) . K^T * scale) V  per head, with LN folded exactly as in st_ln_linear.
extern "C" int st_ln_linear_xattn(const void* x, const float* row_stats, int row_stats_chunks, const void* Wg, const float* c,
                                  const float* d, const void* k, const void* v, void* out, int M, int N, int K, long lda, long ldo,
                                  float eps, int rows_per_batch, int S, int H, long ldk, long ldv, float scale, int dtype,
                                  const void* next_weights, size_t next_weights_bytes, void* stream) {
    ST_REQUIRE(st_dtype_is16(dtype), "ln_linear_xattn: dtype %d not supported (bf16 / f16)", dtype);
    ST_REQUIRE(x && Wg && c && d && k && v && out && row_stats, "ln_linear_xattn: null pointer");
    ST_REQUIRE(row_stats_chunks > 0, "ln_linear_xattn: the producer emitted no row statistics");
    ST_REQUIRE(M > 0 && N > 0 && K > 0 && S > 0 && H > 0, "ln_linear_xattn: bad shape M=%d N=%d K=%d S=%d H=%d", M, N, K, S, H);
    ST_REQUIRE(N == H * 64, "ln_linear_xattn: N=%d must be H*64 (H=%d)", N, H);
    ST_REQUIRE(S < 256, "ln_linear_xattn: context of %d keys (the fused epilogue runs the short-context attention core: S < 256)", S);
    ST_REQUIRE(rows_per_batch > 0 && rows_per_batch % 128 == 0 && M % rows_per_batch == 0,
               "ln_linear_xattn: %d rows per batch: query tiles of 128 rows must not straddle batches", rows_per_batch);
    ST_REQUIRE(K % 64 == 0 && lda % 8 == 0 && ldo % 4 == 0 && ldk % 8 == 0 && ldv % 8 == 0, "ln_linear_xattn: K, strides must keep 16-byte alignment");
    ST_REQUIRE(((uintptr_t)x | (uintptr_t)Wg | (uintptr_t)out | (uintptr_t)k | (uintptr_t)v | (uintptr_t)c | (uintptr_t)d) % 16 == 0,
               "ln_linear_xattn: pointers must be 16-byte aligned");
    GemmArgs a = {};
    a.A = x; a.W = Wg; a.C = out; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldo; a.epi = 0;
    a.ln_c = c; a.ln_d = d; a.ln_eps = eps; a.splitk = 1; a.ln_stats = row_stats; a.ln_chunks = row_stats_chunks;
    a.xa_k = k; a.xa_v = v; a.xa_ldk = ldk; a.xa_ldv = ldv; a.xa_S = S; a.xa_T = rows_per_batch; a.xa_scale_log2e = scale * 1.4426950408889634f;
    take_hint(a, next_weights, next_weights_bytes);
    { void* none; if (int e = st_take_split_arm("ln_linear_xattn", M, N, false, &none)) return e; }
    return dtype == ST_BF16 ? gemm_xattn_bf16(a, (hipStream_t)stream) : gemm_xattn_f16(a, (hipStream_t)stream);
}

// fp8 projections (seed: the reference's fp8-stored projection weights, kernels/attention_proj.py:36-39,
// 105-155, which it up-converts before the product - here both operands go to the fp8 matrix pipe):
//   y = epilogue((xq Wq^T) * row_scale[m] * w_scale[n]),  xq / Wq OCP e4m3 bytes, fp32 accumulation, bf16 out.
extern "C" int st_linear_fp8(const void* xq, const float* row_scale, const void* Wq, const float* w_scale, const void* bias,
                             const void* residual, void* y, int M, int N, int K, long lda, long ldc, long ldr, int epilogue,
                             void* workspace, size_t workspace_bytes, const void* next_weights, size_t next_weights_bytes,
                             void* stream) {
    ST_REQUIRE(xq && row_scale && Wq && w_scale && y, "linear_fp8: null pointer");
    ST_REQUIRE(M > 0 && N > 0 && K > 0, "linear_fp8: bad shape M=%d N=%d K=%d", M, N, K);
    ST_REQUIRE(K % 128 == 0 && lda % 16 == 0, "linear_fp8: K=%d must be a multiple of 128 and lda=%ld of 16", K, lda);
    ST_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && (!(epilogue & ST_EPI_RESIDUAL) || ldr % 4 == 0), "linear_fp8: N, ldc, ldr must be multiples of 4");
    ST_REQUIRE(!(epilogue & ST_EPI_ROWBIAS), "linear_fp8: the row-bias epilogue is not supported");
    ST_REQUIRE(((uintptr_t)xq | (uintptr_t)Wq | (uintptr_t)y | (uintptr_t)w_scale) % 16 == 0, "linear_fp8: pointers must be 16-byte aligned");
    GemmArgs a = {};
    a.A = xq; a.W = Wq; a.bias = bias; a.residual = residual; a.C = y;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldr = ldr; a.epi = epilogue;
    a.row_scale = row_scale; a.col_scale = w_scale; a.rs_stride = 1;
    a.splitk = 1; a.partial = (float*)workspace; a.partial_bytes = workspace ? workspace_bytes : 0;
    take_hint(a, next_weights, next_weights_bytes);
    if (int e = check_epilogue("linear_fp8", a)) return e;
    return gemm_dense_fp8(a, (hipStream_t)stream);
}

// The general fp8 GEMM of the compiled graph's fp8 plan (see the header): activation scale per row or per tensor, optional
// folded LayerNorm, optional e4m3 copy of the output (with or without the bf16 output itself), row statistics for a following
// folded LayerNorm.
extern "C" int st_linear_fp8x(const void* xq, const float* a_scale, int a_scale_stride, const void* Wq, const float* w_scale,
                              const void* bias, const void* residual, void* y, int M, int N, int K, long lda, long ldc, long ldr, int epilogue,
                              const float* ln_stats, int ln_chunks, const float* ln_c, const float* ln_d, float ln_eps,
                              float* row_stats, int row_stats_capacity, int* row_stats_chunks,
                              void* q8, long ldq8, const float* q8_inv_scale, unsigned int* q8_amax,
                              void* workspace, size_t workspace_bytes, const void* next_weights, size_t next_weights_bytes, void* stream) {
    ST_REQUIRE(xq && a_scale && Wq && w_scale && (y || q8), "linear_fp8x: null pointer");
    ST_REQUIRE(a_scale_stride == 0 || a_scale_stride == 1, "linear_fp8x: activation scale stride %d (0 = per tensor, 1 = per row)", a_scale_stride);
    ST_REQUIRE(M > 0 && N > 0 && K > 0, "linear_fp8x: bad shape M=%d N=%d K=%d", M, N, K);
    ST_REQUIRE(K % 128 == 0 && lda % 16 == 0, "linear_fp8x: K=%d must be a multiple of 128 and lda=%ld of 16", K, lda);
    ST_REQUIRE(N % 4 == 0 && (!y || ldc % 4 == 0) && (!(epilogue & ST_EPI_RESIDUAL) || ldr % 4 == 0), "linear_fp8x: N, ldc, ldr must be multiples of 4");
    ST_REQUIRE(!(epilogue & ST_EPI_ROWBIAS), "linear_fp8x: the row-bias epilogue is not supported");
    ST_REQUIRE(((uintptr_t)xq | (uintptr_t)Wq | (uintptr_t)y | (uintptr_t)w_scale) % 16 == 0, "linear_fp8x: pointers must be 16-byte aligned");
    ST_REQUIRE(!ln_c || (ln_d && ln_stats && ln_chunks > 0 && !(epilogue & ~ST_EPI_GEGLU)), "linear_fp8x: folded LayerNorm takes c, d, the row statistics and only the GEGLU flag (bias lives in d)");
    // (only the staged epilogue - every GEGLU tile takes it - honours a missing y; the fragment epilogue of the small tiles stores through it)
    ST_REQUIRE(y || (epilogue & ST_EPI_GEGLU), "linear_fp8x: an output without y (only the e4m3 copy) is the GEGLU form");
    GemmArgs a = {};
    a.A = xq; a.W = Wq; a.bias = bias; a.residual = residual; a.C = y;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldr = ldr; a.epi = epilogue;
    a.row_scale = a_scale; a.col_scale = w_scale; a.rs_stride = a_scale_stride;
    a.ln_c = ln_c; a.ln_d = ln_d; a.ln_eps = ln_eps; a.ln_stats = ln_stats; a.ln_chunks = ln_chunks;
    a.splitk = 1; a.partial = (float*)workspace; a.partial_bytes = workspace ? workspace_bytes : 0;
    ST_REQUIRE(!row_stats || !(epilogue & ST_EPI_GEGLU), "linear_fp8x: row_stats with GEGLU is not supported");
    a.row_stats = row_stats; a.stats_capacity = row_stats_capacity; a.stats_chunks_out = row_stats_chunks;
    take_hint(a, next_weights, next_weights_bytes);
    if (int e = check_epilogue("linear_fp8x", a)) return e;
    if (int e = check_q8("linear_fp8x", a, q8, ldq8, q8_inv_scale, q8_amax)) return e;
    return gemm_dense_fp8(a, (hipStream_t)stream);
}

extern "C" int st_conv2d(const void* x, const void* W, const void* bias, const void* residual, const void* rowbias, void* y,
                         int N, int Hin, int Win, int Cin, int Cout, int R, int S, int stride, int pad, int upsample2x,
                         int epilogue, int dtype, void* workspace, size_t workspace_bytes,
                         float* col_stats, int col_stats_tiles, int* col_stats_rows,
                         const void* next_weights, size_t next_weights_bytes, void* stream) {
    if (col_stats_rows) *col_stats_rows = 0;
    ST_REQUIRE(x && W && y, "conv2d: null pointer");
    ST_REQUIRE(N > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0,
               "conv2d: bad geometry");
    ST_REQUIRE(st_dtype_ok_gemm(dtype), "conv2d: unsupported dtype %d", dtype);
    ST_REQUIRE(!(epilogue & ST_EPI_GEGLU), "conv2d: GEGLU epilogue not supported");
    ST_REQUIRE(Cout % 4 == 0, "conv2d: Cout=%d must be a multiple of 4", Cout);
    const int He = upsample2x ? 2 * Hin : Hin, We = upsample2x ? 2 * Win : Win;
    GemmArgs a = {};
    a.A = x; a.W = W; a.bias = bias; a.residual = residual; a.rowbias = rowbias; a.C = y;
    a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.S = S; a.stride = stride; a.pad = pad; a.ups = upsample2x ? 1 : 0;
    {
        static const int korder_env = dev_env_int("ST_CONV_KORDER", -1);
        a.R_ = R; a.korder = korder_env >= 0 ? korder_env : 0;
    }
    a.Hout = (He + 2 * pad - R) / stride + 1;
    a.Wout = (We + 2 * pad - S) / stride + 1;
    ST_REQUIRE(a.Hout > 0 && a.Wout > 0, "conv2d: empty output");
    a.M = N * a.Hout * a.Wout; a.N = Cout; a.K = R * S * Cin;
    a.lda = 0; a.ldc = Cout; a.ldr = Cout; a.rows_per_batch = a.Hout * a.Wout; a.epi = epilogue;
    a.splitk = 1; a.partial = (float*)workspace; a.partial_bytes = workspace ? workspace_bytes : 0;
    a.col_stats = col_stats; a.col_tiles_cap = col_stats_tiles; a.col_rows_out = col_stats_rows;
    take_hint(a, next_weights, next_weights_bytes);
    if (int e = check_epilogue("conv2d", a)) return e;
    if (int e = st_take_split_arm("conv2d", a.M, Cout, !st_dtype_is16(dtype) && Cout % 32 == 0 && Cin % 32 == 0, &a.sp_out)) return e;
    hipStream_t st = (hipStream_t)stream;
    const int kb = st_dtype_is16(dtype) ? 64 : 32;
    if (Cin % kb == 0) {
        ST_REQUIRE(((uintptr_t)x | (uintptr_t)W | (uintptr_t)y) % 16 == 0, "conv2d: pointers must be 16-byte aligned");
        if (dtype == ST_BF16) return gemm_conv_bf16(a, R, upsample2x, st);
        if (dtype == ST_F16) return gemm_conv_f16(a, R, upsample2x, st);
        if (dtype == ST_F32S) return gemm_conv_f32s(a, R, upsample2x, st);
        return gemm_conv_f32(a, st);
    }
    ST_REQUIRE(dtype != ST_F32S, "conv2d: split fp32 operands need Cin=%d to be a multiple of 32 (thin inputs take plain ST_F32)", Cin);
    // thin-input path: K = R*S*Cin small enough to keep one pixel's inputs in registers
    ST_REQUIRE(a.K <= 64 && Cout % 16 == 0, "conv2d: Cin=%d is neither a multiple of %d (implicit GEMM) nor thin (R*S*Cin <= 64, Cout %% 16 == 0)", Cin, kb);
    return conv_thin_run(a, R, dtype, st);
}

// 1x1 convolution (stride 1, no padding) over the channel concatenation [x0 | x1] without the concatenated tensor: the skip
// connections of the decoder (unet_pt.py:352-357) feed a resnet whose 1x1 shortcut is the only GEMM-shaped reader of the
// concatenation.  Bit-identical to st_conv2d on torch.cat([x0, x1], 1): same K order, same tiles.
extern "C" int st_conv1x1_cat(const void* x0, int C0, const void* x1, int C1, const void* W, const void* bias, const void* residual, void* y,
                              int N, int H, int Wd, int Cout, int epilogue, int dtype, void* workspace, size_t workspace_bytes,
                              float* col_stats, int col_stats_tiles, int* col_stats_rows,
                              const void* next_weights, size_t next_weights_bytes, void* stream) {
    if (col_stats_rows) *col_stats_rows = 0;
    ST_REQUIRE(x0 && x1 && W && y, "conv1x1_cat: null pointer");
    ST_REQUIRE(N > 0 && H > 0 && Wd > 0 && C0 > 0 && C1 > 0 && Cout > 0, "conv1x1_cat: bad geometry");
    ST_REQUIRE(st_dtype_ok_gemm(dtype), "conv1x1_cat: unsupported dtype %d", dtype);
    ST_REQUIRE(!(epilogue & (ST_EPI_GEGLU | ST_EPI_ROWBIAS)), "conv1x1_cat: GEGLU / row-bias epilogues not supported");
    ST_REQUIRE(Cout % 4 == 0, "conv1x1_cat: Cout=%d must be a multiple of 4", Cout);
    const int kb = st_dtype_is16(dtype) ? 64 : 32;
    ST_REQUIRE(C0 % kb == 0 && C1 % kb == 0, "conv1x1_cat: both channel counts (%d, %d) must be multiples of %d", C0, C1, kb);
    ST_REQUIRE(((uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)W | (uintptr_t)y) % 16 == 0, "conv1x1_cat: pointers must be 16-byte aligned");
    GemmArgs a = {};
    a.A = x0; a.A2 = x1; a.Csplit = C0; a.W = W; a.bias = bias; a.residual = residual; a.C = y;
    a.Hin = H; a.Win = Wd; a.Cin = C0 + C1; a.S = 1; a.stride = 1; a.pad = 0; a.ups = 0; a.R_ = 1; a.korder = 0;
    a.Hout = H; a.Wout = Wd;
    a.M = N * H * Wd; a.N = Cout; a.K = C0 + C1;
    a.lda = 0; a.ldc = Cout; a.ldr = Cout; a.rows_per_batch = H * Wd; a.epi = epilogue;
    a.splitk = 1; a.partial = (float*)workspace; a.partial_bytes = workspace ? workspace_bytes : 0;
    a.col_stats = col_stats; a.col_tiles_cap = col_stats_tiles; a.col_rows_out = col_stats_rows;
    take_hint(a, next_weights, next_weights_bytes);
    if (int e = check_epilogue("conv1x1_cat", a)) return e;
    if (int e = st_take_split_arm("conv1x1_cat", a.M, Cout, !st_dtype_is16(dtype) && Cout % 32 == 0, &a.sp_out)) return e;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ST_BF16) return gemm_conv_bf16(a, 1, 0, st);
    if (dtype == ST_F16) return gemm_conv_f16(a, 1, 0, st);
    if (dtype == ST_F32S) return gemm_conv_f32s(a, 1, 0, st);
    return gemm_conv_f32(a, st);
}
