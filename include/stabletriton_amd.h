/*
 * stabletriton_amd - C ABI of the MI355X (gfx950) operator library.
 *
 * This is the drop-in boundary for the SDXL-UNet denoise hot path: each entry
 * point is what the reference's fx leaf wrapper for the same operator would
 * bind instead of its Triton/xformers launch.  Plain pointers and sizes only;
 * every pointer is DEVICE memory unless noted; `stream` is a hipStream_t
 * passed as void*.  Launches are asynchronous on `stream`, never synchronise
 * the host and never allocate, so they are legal inside hipGraph capture
 * (reference requirement: optimizers/cuda/graphs.py:72-108 captures on a side
 * stream).  Return value: 0 = launched, non-zero = rejected before launch
 * (see st_last_error()); the Python host turns non-zero into an exception,
 * mirroring the reference's assert/RuntimeError behaviour
 * (kernels/linear.py:181-188, kernels/geglu.py:29-30).
 */
#ifndef STABLETRITON_AMD_H
#define STABLETRITON_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of activations/weights; accumulation is always fp32.  ST_F16 (IEEE half) is the type the reference's
 * own call site computes in (implementations/Diffusers/load_sdxl_pipeline.py:17-28 passes a .half() module;
 * optimizers/replace_attention.py:91 casts q/k/v to fp16): same matrix-pipe rate as bf16, 10 mantissa bits. */
enum { ST_F32 = 0, ST_BF16 = 1, ST_F16 = 2,
       /* ST_F32S, accepted by the GEMM-shaped entry points only (st_linear, st_ln_linear, st_conv2d, st_conv1x1_cat): the
        * matrix operands x and W are "split fp32" images written by st_split_f32 (or by a producer's epilogue) - every value
        * as two IEEE halves, x ~ hi + lo * 2^-11, 4 bytes per value, row segments of 32 values = [32 hi | 32 lo] - and the
        * product runs as three 16-bit MFMAs per 32 k with fp32 accumulation (22 significant bits per operand); bias,
        * residual, row bias, statistics and y are plain fp32.  The strict (fp32-parity) mode's matrix path. */
       ST_F32S = 3 };

/* activation-tensor layouts for the image-shaped ops */
enum { ST_NCHW = 0, ST_NHWC = 1 };

/* st_linear / st_conv2d epilogue flags (bit-or) */
enum {
    ST_EPI_BIAS      = 1,   /* + bias[n]                                         */
    ST_EPI_SILU      = 2,   /* y = y * sigmoid(y)         (after bias)            */
    ST_EPI_GEGLU     = 4,   /* W has 2F rows; out[m][j] = y[j] * gelu_erf(y[F+j]) */
    ST_EPI_RESIDUAL  = 8,   /* + residual[m][n]           (after activation)      */
    ST_EPI_ROWBIAS   = 16   /* + rowbias[batch(m)][n]     (time-embedding add)    */
};

int         st_abi_version(void);          /* bumps on any signature or contract change; this header is ABI 16
                                              (6: next-weights hint passed per call, st_timestep_sincos; 7: fp8 entry points; 8: GroupNorm partials from the
                                              producer; 9: st_ln_linear_xattn; 10: ST_F16 accepted by every entry point
                                              that takes a dtype, st_ln_linear_xattn takes a dtype; 11: fp8 plan with
                                              delayed per-tensor scaling - st_linear_emit8, st_linear_fp8x, st_fp8_update_scales; 12: readers of a channel
                                              concatenation that is never written - st_group_norm_from_stats_cat, st_conv1x1_cat; 13: ST_F32S split fp32 matrix operands, st_split_f32, st_arm_split_output, st_attention_split; 14: st_attention
                                              takes head_dim 16 / 32 / 128 beside 64; 15: st_timestep_features takes the host's table of the reference's own features for integer timesteps;
                                              16: next_weights_bytes carries the geometry of a strided touch in bits 40-61) */
const char* st_last_error(void);           /* host string, thread-local     */

/* GroupNorm (+SiLU).  Replaces reference group_norm_wrapper
 * (optimizers/replace_groupnorm.py:18-19 -> kernels/groupnorm.py:128-161).
 * x,y: (N,C,H,W) logical, `layout` physical; gamma,beta: C elements of `dtype`;
 * workspace: st_group_norm_workspace_bytes() bytes of scratch.  Statistics are
 * fp32, biased variance, y = (x-mean)*rsqrt(var+eps)*gamma+beta, then
 * y*sigmoid(y) if `silu`. */
size_t st_group_norm_workspace_bytes(int N, int C, int HW, int groups);
int st_group_norm(const void* x, const void* gamma, const void* beta, void* y,
                  int N, int C, int HW, int groups, float eps, int silu,
                  int layout, int dtype, void* workspace, void* stream);

/* GroupNorm (+SiLU) of an NHWC tensor whose statistics come from its producer: `stats0` (and `stats1` for the second
 * half of a channel concatenation, unet_pt.py:352-357; else NULL / 0 / 0) are the `col_stats` buffers of the st_linear /
 * st_conv2d launches that wrote x - C0 (+ C1 = C) channels, rows0 / rows1 rows per partial.  Same arithmetic contract
 * as st_group_norm (fp32 statistics, biased variance; the partial sums are combined in double precision); the statistics
 * pass over x and its launch are gone.  workspace: st_group_norm_workspace_bytes(). */
int st_group_norm_from_stats(const void* x, const void* gamma, const void* beta, void* y, int N, int C, int HW,
                             int groups, float eps, int silu, int dtype, const float* stats0, int C0, int rows0,
                             const float* stats1, int C1, int rows1, void* workspace, void* stream);

/* The same for x = the channel concatenation [x0 | x1] of two NHWC tensors (C0 / C1 = C - C0 channels, multiples of one
 * 16-byte vector), each with the statistics of its own producer; the concatenated tensor is never written (the decoder's
 * skip connections, unet_pt.py:352-357: torch.cat -> norm1).  Bit-identical to st_group_norm_from_stats on the
 * concatenated tensor. */
int st_group_norm_from_stats_cat(const void* x0, const void* x1, const void* gamma, const void* beta, void* y, int N, int C, int HW,
                                 int groups, float eps, int silu, int dtype, const float* stats0, int C0, int rows0,
                                 const float* stats1, int C1, int rows1, void* workspace, void* stream);

/* LayerNorm over the last dimension.  Replaces layer_norm_wrapper
 * (optimizers/replace_layernorm.py:17-24 -> kernels/layer_norm.py:282-335);
 * eps is used as given (the reference's fp16 clamp to 1.6e-5 is a defect,
 * SURVEY.md section 7).  x,y: (rows, C) contiguous. */
int st_layer_norm(const void* x, const void* gamma, const void* beta, void* y,
                  int rows, int C, float eps, int dtype, void* stream);

/* GEGLU elementwise: out[m][j] = state[m][j] * gelu_erf(gate[m][j]).
 * Replaces geglu_triton (optimizers/replace_geglu.py:23-27 ->
 * kernels/geglu.py:18-35).  Row strides are in elements, so the two halves of
 * one projection output can be passed without copies. */
int st_geglu(const void* state, const void* gate, void* out, int rows, int F,
             long ld_state, long ld_gate, long ld_out, int dtype, void* stream);

/* Linear: y[M,N] = epilogue(x[M,K] * W[N,K]^T).  Replaces linear_wrapper /
 * linear_wrapper_functional (optimizers/replace_linear.py:20-34 ->
 * kernels/linear.py:173-222).  W is (N,K) row-major exactly as nn.Linear
 * stores it.  lda/ldc/ldr are row strides in elements.  With ST_EPI_GEGLU,
 * W has 2N rows and y has N columns.  rows_per_batch is only read with
 * ST_EPI_ROWBIAS (rowbias is (M/rows_per_batch, N) contiguous).
 * `workspace` (may be NULL) is caller-owned scratch of `workspace_bytes` bytes that the
 * caller ZEROES ONCE before its first use (hipMemset) and that concurrent launches must
 * not share: when present, long-K problems with few output tiles are split over K inside
 * the one launch - every K slice stores an fp32 slab there, and the block of a tile that
 * finishes last adds the slabs in slice order and applies the epilogue (bit-reproducible).
 * The first 64 KiB hold per-tile arrival counters, which every call leaves at zero again.
 * `row_stats` (may be NULL): device buffer of M * row_stats_capacity float2; when given, the
 * kernel also writes, per output row and per N tile, (sum, sum of squares) of the values it
 * stored - the LayerNorm partials st_ln_linear consumes; the number of tiles actually used is
 * returned through the HOST pointer `row_stats_chunks` (0 = none written).
 * `col_stats` (may be NULL): device buffer of col_stats_tiles * N float2; when given (with rows_per_batch = rows per
 * image), the kernel also writes, per tile row of the launch and per output column, (sum, sum of squares) of the values
 * it stored - the GroupNorm partials st_group_norm_from_stats consumes; the rows per tile row actually used come back
 * through the HOST pointer `col_stats_rows` (0 = none written: tile rows would straddle images, or the shape takes a
 * kernel that cannot emit them).
 * `next_weights` / `next_weights_bytes` (may be NULL / 0; no reference counterpart): the weight matrix the
 * GEMM-shaped launch AFTER this one will read.  This launch touches it (one dword per 128-byte line, spread
 * over its blocks, during its epilogue or from helper blocks on idle CUs) so that it waits in the memory-side
 * cache when its own GEMM starts; without it cold weights cost every GEMM an HBM round trip in its prologue.
 * The launch waits for its touches before it exits - the matrix's bytes at HBM speed -, so a caller hints small matrices whole
 * and large ones STRIDED: `next_weights_bytes` = byte count (bits 0-39) | row length in 128-byte lines << 40 (bits 40-59,
 * 0 = every line) | s << 60 (bits 60-61): the first 2^s lines of every row are touched - the K tiles the next launch's
 * prologue asks for.  The buffer must stay allocated until this launch has run (also under graph replay). */
int st_linear(const void* x, const void* W, const void* bias, const void* residual,
              const void* rowbias, void* y, int M, int N, int K,
              long lda, long ldc, long ldr, int rows_per_batch,
              int epilogue, int dtype, void* workspace, size_t workspace_bytes,
              float* row_stats, int row_stats_capacity, int* row_stats_chunks,
              float* col_stats, int col_stats_tiles, int* col_stats_rows,
              const void* next_weights, size_t next_weights_bytes, void* stream);

/* LayerNorm folded into the Linear (or GEGLU projection) that consumes it - the pair
 * layer_norm_wrapper -> linear_wrapper of the reference graph (replace_layernorm.py:17-24,
 * replace_linear.py:20-34; unet_pt.py:192-208) as ONE launch:
 *   y = rstd_m * (x W'^T - mean_m * c) + d,   W' = W * diag(gamma)  (rows of `Wg`, dtype),
 *   c[n] = sum_k W'[n][k],  d[n] = sum_k beta[k] W[n][k] + bias[n]   (fp32, host-prepared),
 * mean_m / rstd_m = LayerNorm statistics of row m of x, summed from the `row_stats`
 * partials (M x row_stats_chunks float2) the GEMM that produced x emitted (st_linear).
 * With ST_EPI_GEGLU, Wg has 2N rows and c, d 2N entries. */
int st_ln_linear(const void* x, const float* row_stats, int row_stats_chunks, const void* Wg,
                 const float* c, const float* d, void* y, int M, int N, int K, long lda,
                 long ldc, float eps, int epilogue, int dtype,
                 const void* next_weights, size_t next_weights_bytes, void* stream);

/* The query projection of the text-context attention and that attention as ONE launch - the chain
 * layer_norm_wrapper -> linear_wrapper (attn2.to_q) -> attention_wrapper of a transformer block (unet_pt.py:192-208,
 * 133-142; replace_layernorm.py:17-24, replace_linear.py:20-34, replace_attention.py:60-68):
 *   out[M, H*64] = softmax((LN(x) Wq^T + bias) k^T * scale) v   per head,
 * LayerNorm folded as in st_ln_linear (Wg, c, d, row_stats), k / v the (batch, S, H*64) context projections with token
 * strides ldk / ldv (S < 256), rows_per_batch query rows per batch entry (a multiple of 128).  The query tile never
 * leaves the chip: results are bit-identical to st_ln_linear followed by st_attention.  k / v batches are dense
 * (batch stride = S * ldk / S * ldv).  dtype: ST_BF16 or ST_F16. */
int st_ln_linear_xattn(const void* x, const float* row_stats, int row_stats_chunks, const void* Wg,
                       const float* c, const float* d, const void* k, const void* v, void* out,
                       int M, int N, int K, long lda, long ldo, float eps, int rows_per_batch, int S, int H,
                       long ldk, long ldv, float scale, int dtype,
                       const void* next_weights, size_t next_weights_bytes, void* stream);

/* Fused attention core: out = softmax(q k^T * scale) v per head, no mask.
 * Replaces attention_wrapper (optimizers/replace_attention.py:60-68); inputs
 * keep the (B, T, H*D) / (B, S, H*D) projection layout of unet_pt.py:133-142.
 * ld* are token strides in elements (>= H*D), batch strides are T*ldq etc.
 * D (head_dim) in {16, 32, 64, 128}, the reference operator's set (kernels/attention_fa2.py:118-123); 64 - every SDXL
 * head - runs on the tuned kernels, the other three on one generic kernel (csrc/attention_anyd.hip). */
int st_attention(const void* q, const void* k, const void* v, void* out,
                 int B, int T, int S, int H, int D,
                 long ldq, long ldk, long ldv, long ldo,
                 float scale, int dtype, void* stream);

/* conv2d on NHWC activations as implicit GEMM; W is (Cout, R, S, Cin)
 * contiguous (= channels_last nn.Conv2d weight).  Covers the resnet-block
 * convolutions of unet_pt.py:74-95,246-266,430,467 that the reference leaves
 * to cuDNN (optimizations.txt:5).  `upsample2x` folds a nearest 2x upsample of
 * the input into the gather (unet_pt.py:264-266).  Epilogue flags as for
 * st_linear; rowbias is (N_batch, Cout) (the time-embedding projection,
 * unet_pt.py:82-83), residual is NHWC (N,Hout,Wout,Cout).  workspace, col_stats, next_weights: as st_linear
 * (rows per image = Hout*Wout). */
int st_conv2d(const void* x, const void* W, const void* bias, const void* residual,
              const void* rowbias, void* y, int N, int Hin, int Win, int Cin,
              int Cout, int R, int S, int stride, int pad, int upsample2x,
              int epilogue, int dtype, void* workspace, size_t workspace_bytes,
              float* col_stats, int col_stats_tiles, int* col_stats_rows,
              const void* next_weights, size_t next_weights_bytes, void* stream);

/* 1x1 convolution (stride 1, no padding) of the channel concatenation [x0 | x1] (NHWC, C0 / C1 channels, multiples of a
 * K tile: 64 for 16-bit types, 32 for fp32) without the concatenated tensor: the resnet shortcut behind a skip connection
 * (unet_pt.py:352-357 -> 74-95).  W is (Cout, 1, 1, C0 + C1).  Epilogue flags BIAS / SILU / RESIDUAL; workspace, col_stats,
 * next_weights as st_conv2d.  Bit-identical to st_conv2d on the concatenated tensor. */
int st_conv1x1_cat(const void* x0, int C0, const void* x1, int C1, const void* W, const void* bias, const void* residual, void* y,
                   int N, int H, int Wd, int Cout, int epilogue, int dtype, void* workspace, size_t workspace_bytes,
                   float* col_stats, int col_stats_tiles, int* col_stats_rows,
                   const void* next_weights, size_t next_weights_bytes, void* stream);

/* Euler-discrete update of the fp32 latent and preparation of the next UNet
 * input (restated diffusers EulerDiscreteScheduler, see
 * stabletriton_amd/scheduler.py; the reference leaves this loop to the
 * third-party pipeline, implementations/Diffusers/load_sdxl_pipeline.py:39-46):
 *   i = *step;  latent += eps * dsigma[i];  next_in = latent * in_scale[i+1]
 * (cast to `dtype`).  All three tensors are elementwise-aligned (same layout),
 * n elements.  `step` lives on the device so one captured step graph can be
 * replayed down the table; st_step_advance does *step = (*step + 1) % n_steps. */
int st_euler_step(float* latent, const void* eps, void* next_in, const float* dsigma,
                  const float* in_scale, const int* step, long n, int n_steps,
                  int dtype, void* stream);
int st_step_advance(int* step, int n_steps, void* stream);

/* Sinusoidal timestep features (unet_pt.py:17-36; target of the reference's
 * fuse_timesteps pass, optimizers/replace_timesteps.py:33-58):
 *   out[b][j] = cos(t_b * f_j), out[b][dim/2 + j] = sin(t_b * f_j),
 *   f_j = exp(-ln(1e4) * j / (dim/2)),  t_b = t[(step ? *step : 0) + b*t_stride].
 * t is fp32 on the device; out is (batch, dim) of `dtype`.
 * `table` (optional, device, fp32, table_rows x dim): row i = the features of t = i as the REFERENCE's eager path computes
 * them (the host fills it with the reference's own op sequence).  The function is ill-conditioned - t_b * f_j reaches 1e3 rad,
 * so one ulp of exp() moves a feature by 1.2e-4 and two correct fp32 implementations disagree by that much; a timestep that
 * is an integer in [0, table_rows) (every entry of SDXL's schedules, every size / crop of time_ids) therefore takes its row,
 * anything else is computed.  NULL: always computed. */
int st_timestep_features(const float* t, long t_stride, const int* step, void* out,
                         int batch, int dim, int dtype, const float* table, int table_rows, void* stream);

/* ---- fp8 projection path (BASELINE config #5).  Seed in the reference: fp8-stored projection
 * weights, up-converted before the product (kernels/attention_proj.py:36-39, 105-155); here both operands stay OCP
 * e4m3 ("e4m3fn") down to the matrix pipe - the block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 at twice the bf16 rate, its
 * E8M0 block scales all 2^0 (the scales of this path are per row / per output channel, applied in the epilogue) -
 * accumulation fp32, output bf16.
 *
 * st_quantize_fp8: x (rows, C) of `dtype` (any of the three), row stride ldx elements -> xq (rows, C) e4m3 bytes, contiguous, and
 *   row_scale[m] = max_k |x[m][k]| / 448 (fp32), xq[m][k] = e4m3(x[m][k] / row_scale[m]), round to nearest even.
 * st_layer_norm_quantize_fp8: the same on LayerNorm(x) (the layer_norm_wrapper -> linear_wrapper pair of the
 *   transformer blocks as one pass over x); x (rows, C) contiguous.
 * st_linear_fp8: y[M,N] = epilogue((xq Wq^T) * row_scale[m] * w_scale[n]); Wq is (N, K) e4m3 bytes (2N rows and 2N
 *   scales with ST_EPI_GEGLU), K a multiple of 128, bias / residual / y bf16.  workspace, next_weights: as st_linear. */
int st_quantize_fp8(const void* x, long ldx, void* xq, float* row_scale, int rows, int C, int dtype, void* stream);
int st_layer_norm_quantize_fp8(const void* x, const void* gamma, const void* beta, void* xq, float* row_scale,
                               int rows, int C, float eps, int dtype, void* stream);
int st_linear_fp8(const void* xq, const float* row_scale, const void* Wq, const float* w_scale, const void* bias,
                  const void* residual, void* y, int M, int N, int K, long lda, long ldc, long ldr, int epilogue,
                  void* workspace, size_t workspace_bytes, const void* next_weights, size_t next_weights_bytes,
                  void* stream);

/* ---- fp8 plan of the compiled graph (optimizers/plan_fp8.py): the three big projections of a transformer block (q|k|v, the
 * GEGLU projection, the feed-forward output) run with e4m3 operands and NO quantisation launches.  The launch that produces a
 * projection's input also writes an e4m3 copy of it, scaled by a PER-TENSOR factor derived from the previous denoise step's
 * max |value| ("delayed scaling"); this step's maximum goes to the tensor's `amax` partial slots (256 unsigned ints holding
 * non-negative float bit patterns, combined by atomic max).  st_fp8_update_scales runs once per step before the first launch:
 * for tensor i, scale[i] = margin * max(amax_parts[i][0..255]) / 448, inv_scale[i] = 1 / scale[i], partials cleared; a tensor
 * whose partials are all zero keeps its scale.
 *
 * st_linear_emit8: st_linear whose epilogue also writes q8[m][n] = e4m3(clamp(y[m][n] * *q8_inv_scale, +-448)) (row stride ldq8
 *   bytes, N and ldq8 multiples of 8) and this launch's max |y| to q8_amax.  16-bit dtypes.
 * st_linear_fp8x: y[M,N] = epilogue((xq Wq^T) * a_scale[m * a_scale_stride] * w_scale[n]) - a_scale_stride 0: one scale for the
 *   whole activation tensor (the scale[i] above), 1: per row (st_quantize_fp8's row_scale).  With ln_c / ln_d / ln_stats the
 *   LayerNorm in front of the projection is folded exactly as in st_ln_linear (xq is then the e4m3 copy of the UN-normalised
 *   input, c[n] = sum_k of the dequantised folded weights).  `y` may be NULL, with ST_EPI_GEGLU only, when just the e4m3 copy q8 of the
 *   output is wanted (the GEGLU projection feeding the feed-forward output projection).  row_stats as in st_linear. */
int st_linear_emit8(const void* x, const void* W, const void* bias, const void* residual,
                    const void* rowbias, void* y, int M, int N, int K,
                    long lda, long ldc, long ldr, int rows_per_batch,
                    int epilogue, int dtype, void* workspace, size_t workspace_bytes,
                    float* row_stats, int row_stats_capacity, int* row_stats_chunks,
                    float* col_stats, int col_stats_tiles, int* col_stats_rows,
                    void* q8, long ldq8, const float* q8_inv_scale, unsigned int* q8_amax,
                    const void* next_weights, size_t next_weights_bytes, void* stream);
int st_linear_fp8x(const void* xq, const float* a_scale, int a_scale_stride, const void* Wq, const float* w_scale,
                   const void* bias, const void* residual, void* y, int M, int N, int K, long lda, long ldc, long ldr, int epilogue,
                   const float* ln_stats, int ln_chunks, const float* ln_c, const float* ln_d, float ln_eps,
                   float* row_stats, int row_stats_capacity, int* row_stats_chunks,
                   void* q8, long ldq8, const float* q8_inv_scale, unsigned int* q8_amax,
                   void* workspace, size_t workspace_bytes, const void* next_weights, size_t next_weights_bytes, void* stream);
int st_fp8_update_scales(float* scale, float* inv_scale, unsigned int* amax_parts, int n_tensors, float margin, void* stream);

/* Split fp32 images (ST_F32S above; the matrix operands of the strict mode): x (rows, K) fp32, row stride ldx elements ->
 * xs (rows, K) contiguous, 4 bytes per value: per row and per group of 32 consecutive k one 128-byte segment, bytes [0, 64)
 * hi[k] = f16(x[k]), bytes [64, 128) lo[k] = f16((x[k] - hi[k]) * 2048).  K % 32 == 0.  No reference counterpart: the
 * reference's strict path is torch eager fp32 (optimizers/unet_pt.py:469-542). */
int st_split_f32(const float* x, void* xs, long rows, int K, long ldx, void* stream);
/* Split image from the PRODUCER: arms the next launch on the calling thread - one of st_linear, st_ln_linear, st_conv2d,
 * st_conv1x1_cat, st_group_norm, st_group_norm_from_stats[_cat], st_attention with fp32 outputs (ST_F32 / ST_F32S) - to
 * write, beside its output y of (rows, cols) values (cols % 32 == 0; rows = M, pixels or (batch, token)), the split image
 * of y to ys (rows * cols * 4 bytes), which a following GEMM-shaped launch takes as its ST_F32S operand: no st_split_f32
 * launch, no second read of y.  That launch disarms it.  An armed launch that cannot emit (16-bit element type, other
 * shape) is rejected.  ys == NULL disarms: a caller whose armed launch failed its own argument checks (which run before the arm
 * is looked at) disarms before it frees the image, so that no later launch can write to it.  Thread-local, like st_last_error(). */
int st_arm_split_output(void* ys, long rows, int cols);
/* st_attention (ST_F32) whose K and V the producer left as split images: ks / vs point at row 0, first column of head 0, of the
 * image(s) (rows = B * S), k_cols / v_cols = values per image row (the fused q|k|v projection's image has 3 * H * D);
 * q (B, T, ldq) and out (B, T, ldo) plain fp32.  Same results as st_attention, which splits K / V tiles itself. */
int st_attention_split(const void* q, const void* ks, const void* vs, void* out, int B, int T, int S, int H, int D,
                       long ldq, long k_cols, long v_cols, long ldo, float scale, void* stream);

/* The reference's own timestep operator, elementwise (optimizers/replace_timesteps.py:33-40 ->
 * kernels/timestep.py:13-45): x is fp32 of shape (..., half), n elements in all;
 *   sin_out[i] = sin(x[i] * f_j), cos_out[i] = cos(x[i] * f_j), j = i % half, f_j = exp(-ln(1e4) * j / half). */
int st_timestep_sincos(const float* x, float* sin_out, float* cos_out, long n, int half, void* stream);

#ifdef __cplusplus
}
#endif
#endif
