// a8: stand-alone positional encoding (PositionalEncoder.encode used on its own).
// The rendering path never calls this: there the encoding is produced in registers
// inside the fused MLP kernel (mlp_forward.hip) and never touches HBM.
// One thread per output element: 4 B written per thread, fully coalesced.
#include "common.h"

namespace {

__global__ void posenc_kernel(const float *__restrict__ x, int64_t M, int C, int L, int include_input,
                              float *__restrict__ out) {
    const int E = 2 * L * C + (include_input ? C : 0);
    const int64_t total = M * E;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = g / E;
        int e = (int)(g - m * E);
        float v;
        if (include_input && e < C) {
            v = x[m * C + e];  // positional_encoder.py:83-84
        } else {
            if (include_input) e -= C;
            const int f = e / (2 * C);          // frequency level
            const int r = e - f * 2 * C;        // [sin(C channels), cos(C channels)]
            const int c = r < C ? r : r - C;
            const float arg = ldexpf(x[m * C + c], f);  // 2^f * x, exact (:81, :87-88)
            v = r < C ? sinf(arg) : cosf(arg);
        }
        out[g] = v;
    }
}

}  // namespace

NERF_API int nerf_posenc(const float *x, int64_t M, int C, int L, int include_input, float *out,
                         nerf_stream_t stream) {
    NERF_REQUIRE(M >= 0 && C > 0 && L >= 0 && L < 64, "nerf_posenc: bad sizes");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(x && out, "nerf_posenc: null pointer");
    const int E = 2 * L * C + (include_input ? C : 0);
    if (E == 0) return NERF_OK;
    const int64_t total = M * E;
    const int block = 256;
    int64_t grid = (total + block - 1) / block;
    if (grid > 256 * 32) grid = 256 * 32;  // grid-stride beyond 32 blocks per CU
    hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)grid), dim3(block), 0, nerf::as_stream(stream), x, M,
                       C, L, include_input, out);
    return nerf::check_launch("nerf_posenc");
}
