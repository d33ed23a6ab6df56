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
//
// The family lives in one header per kernel: gemm_args.h (descriptor, instruction traits, tile map), epilogue.h,
// gemm_reg.h (register-staged, ragged K), gemm_dma.h (LDS-DMA kernel, split-K combine), gemm8p.h (eight-phase 256-row
// kernel), conv_halo.h, conv_thin.h, dispatch.h (tile configurations and the cost model).  This file includes them all.
#pragma once
#include "gemm_args.h"
#include "epilogue.h"
#include "gemm_reg.h"
#include "gemm_dma.h"
#include "gemm8p.h"
#include "conv_halo.h"
#include "conv_thin.h"
#include "dispatch.h"
