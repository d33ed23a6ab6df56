// The four-wave 256 x 256 GEMM (gemm4w.h) for every element type, in a translation unit of its own (compile time).
// (With the build's -amdgpu-mfma-vgpr-form the 256 accumulator registers take the VGPR half of the register file and hipcc
//  reads the fragments straight into AGPRs; without the flag it shuffles accumulators between the halves and spills.)
#include "../../stabletriton_amd/csrc/gemm_core.h"

// Developer kernel (not part of the product library): built by tools/build_variant.sh / build_one_variant.sh with -DST_DEV_CONFIGS.
#ifdef ST_DEV_CONFIGS

void gemm4w_bf16(const GemmArgs& a, hipStream_t st) { gemm4w_launch<bf16>(a, st); }
#ifndef ST_4W_BF16_ONLY      // (developer builds of one element type compile in a third of the time)
void gemm4w_f16(const GemmArgs& a, hipStream_t st) { gemm4w_launch<f16>(a, st); }
void gemm4w_fp8(const GemmArgs& a, hipStream_t st) { gemm4w_launch<f8>(a, st); }
#endif
#endif      // ST_DEV_CONFIGS
