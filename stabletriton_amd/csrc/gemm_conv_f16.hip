// One element type of the GEMM-shaped kernels (see gemm_core.h, "per-element-type runners").
#include "gemm_core.h"

int gemm_conv_f16(const GemmArgs& a, int R, int ups, hipStream_t st) {
    if (conv_halo_applies(a, R, ups)) return conv_halo_launch<f16>(a, st);
    return gemm_dispatch<f16, true>(a, st);
}
