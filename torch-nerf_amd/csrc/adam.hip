// f1: one torch.optim.Adam step over a flat parameter blob (both networks at once).
//
// The reference builds Adam(params, lr=init_lr, eps=eps) with default betas, no weight decay, no
// amsgrad (runners/runner_utils.py:691-695) and steps it once per batch (runners/train.py:216).
// torch applies it as ~10 multi-tensor launches over 44 tensors; here it is one launch over one
// blob.  HBM-bound: 16 B read + 12 B written per parameter (33 MB for two networks).
//
// Arithmetic follows torch's single-tensor formulation (torch/optim/adam.py, _single_tensor_adam),
// every scalar cast to fp32 the way ATen casts a Python float applied to an fp32 tensor:
//   m  = m + (1-b1) * (g - m)                       exp_avg.lerp_(grad, 1 - beta1)
//   v  = v*b2 + ((1-b2) * g) * g                    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
//   d  = sqrt(v) / sqrt(1 - b2^t) + eps             (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
//   p  = p + (-(lr / (1 - b1^t))) * (m / d)         param.addcdiv_(exp_avg, denom, value=-step_size)
// The bias corrections and step size are computed on the host in double, as torch does.
#include <math.h>

#include "common.h"

namespace {

struct AdamScalars {
    float one_minus_b1, b2, one_minus_b2, bc2_sqrt, eps, neg_step_size, grad_scale;
};

__device__ __forceinline__ void adam_update(float &p, float g, float &m, float &v, const AdamScalars &k) {
    g *= k.grad_scale;
    m = fmaf(k.one_minus_b1, g - m, m);   // ATen's lerp is a fused multiply-add for |weight| < 0.5
    v = v * k.b2 + (k.one_minus_b2 * g) * g;
    const float denom = sqrtf(v) / k.bc2_sqrt + k.eps;
    p = p + k.neg_step_size * (m / denom);
}

// The four blobs share one 16-byte phase: `head` scalars bring them to a float4 boundary.
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ P, const float *__restrict__ G,
                                                   float *__restrict__ M, float *__restrict__ V, int64_t n,
                                                   int head, AdamScalars k) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = (n - head) >> 2;
    float4 *P4 = reinterpret_cast<float4 *>(P + head);
    const float4 *G4 = reinterpret_cast<const float4 *>(G + head);
    float4 *M4 = reinterpret_cast<float4 *>(M + head);
    float4 *V4 = reinterpret_cast<float4 *>(V + head);
    for (int64_t i = tid; i < n4; i += stride) {
        float4 p = P4[i], m = M4[i], v = V4[i];
        const float4 g = G4[i];
        adam_update(p.x, g.x, m.x, v.x, k);
        adam_update(p.y, g.y, m.y, v.y, k);
        adam_update(p.z, g.z, m.z, v.z, k);
        adam_update(p.w, g.w, m.w, v.w, k);
        P4[i] = p;
        M4[i] = m;
        V4[i] = v;
    }
    // ragged ends: up to 3 scalars before the first float4 and up to 3 after the last
    if (tid < head) adam_update(P[tid], G[tid], M[tid], V[tid], k);
    const int64_t t = head + (n4 << 2) + tid;
    if (t < n) adam_update(P[t], G[t], M[t], V[t], k);
}

}  // namespace

NERF_API int nerf_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                            int64_t step, double lr, double beta1, double beta2, double eps, double grad_scale,
                            nerf_stream_t stream) {
    NERF_REQUIRE(n >= 0, "nerf_adam_step: n < 0");
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq, "nerf_adam_step: null pointer");
    NERF_REQUIRE(step >= 1, "nerf_adam_step: step counts from 1");
    NERF_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && lr >= 0.0,
                 "nerf_adam_step: lr/betas/eps out of range");
    const uintptr_t phase = (uintptr_t)params & 15;
    NERF_REQUIRE((phase & 3) == 0 && ((uintptr_t)grads & 15) == phase && ((uintptr_t)exp_avg & 15) == phase &&
                     ((uintptr_t)exp_avg_sq & 15) == phase,
                 "nerf_adam_step: the four blobs must share one 16-byte phase");
    int64_t head = ((16 - phase) & 15) >> 2;
    if (head > n) head = n;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars k;
    k.one_minus_b1 = (float)(1.0 - beta1);
    k.b2 = (float)beta2;
    k.one_minus_b2 = (float)(1.0 - beta2);
    k.bc2_sqrt = (float)sqrt(bc2);
    k.eps = (float)eps;
    k.neg_step_size = (float)(-(lr / bc1));
    k.grad_scale = (float)grad_scale;
    const int64_t n4 = (n + 3) >> 2;   // >= 1, so the grid always covers both ragged ends
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, nerf::as_stream(stream), params,
                       grads, exp_avg, exp_avg_sq, n, (int)head, k);
    return nerf::check_launch("nerf_adam_step");
}
