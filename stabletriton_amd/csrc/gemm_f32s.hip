// One element type of the GEMM-shaped kernels (see gemm_core.h, "per-element-type runners"): split fp32 operands (ST_F32S).
#include "gemm_core.h"

int gemm_dense_f32s(const GemmArgs& a, hipStream_t st) { return gemm_dispatch<fsp, false>(a, st); }
int gemm_conv_f32s(const GemmArgs& a, int R, int ups, hipStream_t st) {
    if (conv_halo_applies(a, R, ups, 32)) return conv_halo_launch<fsp>(a, st);
    return gemm_dispatch<fsp, true>(a, st);
}
