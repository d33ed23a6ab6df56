// One element type of the GEMM-shaped kernels (see gemm_core.h, "per-element-type runners").
#include "gemm_core.h"

// strict fp32 parity mode (one tile configuration) and the thin-input direct conv of all three element types
int gemm_dense_f32(const GemmArgs& a, hipStream_t st) { return gemm_dispatch<float, false>(a, st); }
int gemm_conv_f32(const GemmArgs& a, hipStream_t st) { return gemm_dispatch<float, true>(a, st); }

int conv_thin_run(const GemmArgs& a, int R, int dtype, hipStream_t st) {
    switch (dtype) {
        case ST_BF16: return conv_thin_launch<bf16>(a, R, st);
        case ST_F16: return conv_thin_launch<f16>(a, R, st);
        default: return conv_thin_launch<float>(a, R, st);
    }
}
