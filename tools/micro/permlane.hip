// prints what v_permlane32_swap / v_permlane16_swap return for x = lane id (developer check)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    unsigned x = threadIdx.x;
    u2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    u2 q = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    o[threadIdx.x * 4 + 0] = r[0]; o[threadIdx.x * 4 + 1] = r[1]; o[threadIdx.x * 4 + 2] = q[0]; o[threadIdx.x * 4 + 3] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 16); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) printf("lane %2d: swap32 = (%2u, %2u)  swap16 = (%2u, %2u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
