// One element type of the GEMM-shaped kernels (see gemm_core.h, "per-element-type runners").
#include "gemm_core.h"

int gemm_dense_fp8(const GemmArgs& a, hipStream_t st) { return gemm_dispatch<f8, false>(a, st); }
