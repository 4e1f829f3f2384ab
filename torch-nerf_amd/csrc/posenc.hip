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

// reverse of the above for one input element per thread: d/dx sin(2^f x) = 2^f cos(2^f x), d/dx cos(2^f x) = -2^f sin(2^f x)
// (what autograd accumulates for `in_signal` through torch.cat([fn(in_signal) ...]), positional_encoder.py:104);
// terms are added in the order of the concatenation, like autograd's accumulation of the slices
__global__ void posenc_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g_out, int64_t M, int C, int L,
                                  int include_input, float *__restrict__ g_x) {
    const int E = 2 * L * C + (include_input ? C : 0);
    const int64_t total = M * C;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = g / C;
        const int c = (int)(g - m * C);
        const float *go = g_out + m * E;
        const float v = x[g];
        float acc = include_input ? go[c] : 0.0f;
        const int base = include_input ? C : 0;
        for (int f = 0; f < L; ++f) {
            const float arg = ldexpf(v, f), scale = ldexpf(1.0f, f);
            acc = acc + go[base + 2 * C * f + c] * (scale * cosf(arg));
            acc = acc + go[base + 2 * C * f + C + c] * (-(scale * sinf(arg)));
        }
        g_x[g] = acc;
    }
}

}  // namespace

NERF_API int nerf_posenc_backward(const float *x, const float *g_out, int64_t M, int C, int L, int include_input,
                                  float *g_x, nerf_stream_t stream) {
    NERF_REQUIRE(M >= 0 && C > 0 && L >= 0 && L < 64, "nerf_posenc_backward: bad sizes");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(x && g_out && g_x, "nerf_posenc_backward: null pointer");
    const int64_t total = M * C;
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(posenc_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, nerf::as_stream(stream), x, g_out, M, C, L,
                       include_input, g_x);
    return nerf::check_launch("nerf_posenc_backward");
}

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
