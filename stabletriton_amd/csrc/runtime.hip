// Error plumbing, ABI version, and the small device-side pieces of the denoise
// loop (Euler update, step counter, sinusoidal timestep features).
#include "common.h"

static thread_local char g_err[512] = "";

// ---- split image of the NEXT launch's output (strict mode; header: st_arm_split_output) -------------------------------
static thread_local struct { void* p; long rows; int cols; } g_split_arm = {nullptr, 0, 0};

extern "C" int st_arm_split_output(void* ys, long rows, int cols) {
    if (!ys) { g_split_arm = {nullptr, 0, 0}; return 0; }      // disarm (a caller whose armed launch was rejected before it looked at the arm)
    ST_REQUIRE(rows > 0 && cols > 0 && cols % 32 == 0 && (uintptr_t)ys % 16 == 0, "arm_split_output: (rows, cols) image with cols %% 32 == 0 expected");
    g_split_arm = {ys, rows, cols};
    return 0;
}

// Called by every entry point that can emit: returns the armed image for an output of (rows, cols) fp32 values and disarms
// it; a launch that is armed but cannot emit (other element type, other shape) fails, so an armed image is never left
// unwritten without the caller hearing of it.
int st_take_split_arm(const char* who, long rows, int cols, bool can_emit, void** out) {
    *out = nullptr;
    if (!g_split_arm.p) return 0;
    const auto arm = g_split_arm;
    g_split_arm = {nullptr, 0, 0};
    ST_REQUIRE(can_emit, "%s: a split output image was armed, but this launch cannot emit one (fp32 outputs only)", who);
    ST_REQUIRE(arm.rows == rows && arm.cols == cols, "%s: the armed split image is (%ld, %d), the output is (%ld, %d)", who, arm.rows, arm.cols, rows, cols);
    *out = arm.p;
    return 0;
}

int st_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

int st_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return st_fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

extern "C" const char* st_last_error(void) { return g_err; }
extern "C" int st_abi_version(void) { return 16; }

// ---- Euler-discrete update ---------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void euler_kernel(float* __restrict__ latent, const T* __restrict__ eps, T* __restrict__ next_in,
                                                    const float* __restrict__ dsigma, const float* __restrict__ in_scale,
                                                    const int* __restrict__ step, long n, int n_steps) {
    const int i = *step;
    const float ds = dsigma[i];
    const float sc = in_scale[i + 1 < n_steps ? i + 1 : n_steps - 1];
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long)gridDim.x * 256) {
        float x = latent[j] + Elem<T>::to_f(eps[j]) * ds;
        latent[j] = x;
        next_in[j] = Elem<T>::from_f(x * sc);
    }
}

extern "C" int st_euler_step(float* latent, const void* eps, void* next_in, const float* dsigma, const float* in_scale,
                             const int* step, long n, int n_steps, int dtype, void* stream) {
    ST_REQUIRE(latent && eps && next_in && dsigma && in_scale && step, "euler_step: null pointer");
    ST_REQUIRE(n > 0 && n_steps > 0, "euler_step: bad sizes");
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ST_BF16)
        hipLaunchKernelGGL(euler_kernel<bf16>, dim3(grid), dim3(256), 0, st, latent, (const bf16*)eps, (bf16*)next_in, dsigma, in_scale, step, n, n_steps);
    else if (dtype == ST_F16)
        hipLaunchKernelGGL(euler_kernel<f16>, dim3(grid), dim3(256), 0, st, latent, (const f16*)eps, (f16*)next_in, dsigma, in_scale, step, n, n_steps);
    else if (dtype == ST_F32)
        hipLaunchKernelGGL(euler_kernel<float>, dim3(grid), dim3(256), 0, st, latent, (const float*)eps, (float*)next_in, dsigma, in_scale, step, n, n_steps);
    else
        return st_fail("euler_step: unsupported dtype %d", dtype);
    return st_check_launch("euler_step");
}

__global__ void step_advance_kernel(int* step, int n_steps) {
    int s = *step + 1;
    *step = s >= n_steps ? 0 : s;
}

extern "C" int st_step_advance(int* step, int n_steps, void* stream) {
    ST_REQUIRE(step && n_steps > 0, "step_advance: bad arguments");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, n_steps);
    return st_check_launch("step_advance");
}

// ---- sinusoidal features (reference unet_pt.py:17-36; its fuse_timesteps pass,
// optimizers/replace_timesteps.py:43-58, targets the same sub-graph) -----------
template <typename T>
__global__ void timestep_kernel(const float* __restrict__ t, long t_stride, const int* __restrict__ step,
                                T* __restrict__ out, int batch, int dim, const float* __restrict__ table, int table_rows) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * half) return;
    const int b = idx / half, j = idx - b * half;
    const int base = step ? *step : 0;
    const float tv = t[base + b * t_stride];
    // an integer timestep inside the host's table: the reference's own bits (the header says why)
    if (table && tv >= 0.f && tv < (float)table_rows && tv == floorf(tv)) {
        const float* row = table + (size_t)(int)tv * dim;
        out[(size_t)b * dim + j] = Elem<T>::from_f(row[j]);
        out[(size_t)b * dim + half + j] = Elem<T>::from_f(row[half + j]);
        return;
    }
    // same fp32 op order as the eager module: (-ln(1e4) * j) / half, exp, * t
    const float e = (-9.210340371976184f * (float)j) / (float)half;
    const float a = tv * expf(e);
    out[(size_t)b * dim + j] = Elem<T>::from_f(cosf(a));
    out[(size_t)b * dim + half + j] = Elem<T>::from_f(sinf(a));
}

extern "C" int st_timestep_features(const float* t, long t_stride, const int* step, void* out, int batch, int dim,
                                    int dtype, const float* table, int table_rows, void* stream) {
    ST_REQUIRE(t && out && batch > 0 && dim > 0 && dim % 2 == 0, "timestep_features: bad arguments");
    ST_REQUIRE(!table || table_rows > 0, "timestep_features: a table of %d rows", table_rows);
    if (!table) table_rows = 0;
    const int n = batch * (dim / 2);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ST_BF16)
        hipLaunchKernelGGL(timestep_kernel<bf16>, dim3(cdiv(n, 256)), dim3(256), 0, st, t, t_stride, step, (bf16*)out, batch, dim, table, table_rows);
    else if (dtype == ST_F16)
        hipLaunchKernelGGL(timestep_kernel<f16>, dim3(cdiv(n, 256)), dim3(256), 0, st, t, t_stride, step, (f16*)out, batch, dim, table, table_rows);
    else if (dtype == ST_F32)
        hipLaunchKernelGGL(timestep_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, t, t_stride, step, (float*)out, batch, dim, table, table_rows);
    else
        return st_fail("timestep_features: unsupported dtype %d", dtype);
    return st_check_launch("timestep_features");
}

// ---- the reference's own timestep operator (optimizers/replace_timesteps.py:33-40 -> kernels/timestep.py:13-45):
// elementwise over an already broadcast tensor x of shape (..., half):
//   sin_out[i] = sin(x[i] * f_j), cos_out[i] = cos(x[i] * f_j),  j = i % half,  f_j = exp(-ln(1e4) * j / half)
__global__ void timestep_sincos_kernel(const float* __restrict__ x, float* __restrict__ s, float* __restrict__ c, long n, int half) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = (int)(i % half);
    const float e = (-9.210340371976184f * (float)j) / (float)half;
    const float a = x[i] * expf(e);
    s[i] = sinf(a);
    c[i] = cosf(a);
}

extern "C" int st_timestep_sincos(const float* x, float* sin_out, float* cos_out, long n, int half, void* stream) {
    ST_REQUIRE(x && sin_out && cos_out && n > 0 && half > 0, "timestep_sincos: bad arguments");
    hipLaunchKernelGGL(timestep_sincos_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, sin_out, cos_out, n, half);
    return st_check_launch("timestep_sincos");
}
