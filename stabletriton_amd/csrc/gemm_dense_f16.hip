// One element type of the GEMM-shaped kernels (see gemm_core.h, "per-element-type runners").
#include "gemm_core.h"

int gemm_dense_f16(const GemmArgs& a, hipStream_t st) { return gemm_dispatch<f16, false>(a, st); }

int gemm_xattn_f16(const GemmArgs& a, hipStream_t st) {
    launch_dma_one<f16, 128, 64, 4, 2, 4, 1, false, false, true, true>(a, st, a.N / 64);
    return st_check_launch("ln_linear_xattn");
}
