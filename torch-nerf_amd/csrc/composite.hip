// a11 / a13: the quadrature volume-rendering integral and its reverse.
//
// One wavefront per ray, 64 samples per step, lane = sample.  The exclusive prefix
// sum of sigma*delta is a wave-level scan done with lane shuffles in double (ATen's
// CPU cumsum accumulates in double), carried across 64-sample steps.  HBM-bound:
// forward reads sigma 4 + radiance 12 + delta 4 and writes w 4 bytes per sample
// (+12 B/ray); backward reads 20 (+4 with g_w) and writes 16.
#include "common.h"
#include "render_device.h"

namespace {

using render::wave_inclusive_scan;
using render::wave_sum;

// inclusive scan from the high lane downwards: out[l] = sum_{k >= l} v[k]
__device__ __forceinline__ double wave_inclusive_scan_rev(double v, int lane) {
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const double dn = __shfl_down(v, off, WAVE);
        if (lane + off < WAVE) v += dn;
    }
    return v;
}

constexpr int RAYS_PER_BLOCK = 4;  // 4 independent waves per workgroup

__global__ __launch_bounds__(RAYS_PER_BLOCK *WAVE) void composite_fwd_kernel(
    const float *__restrict__ sigma, const float *__restrict__ radiance,
    const float *__restrict__ delta, int64_t n, int S, float *__restrict__ rgb,
    float *__restrict__ weights) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n) return;  // whole wave exits together
    const float *sg = sigma + ray * S;
    const float *dl = delta + ray * S;
    const float *cl = radiance + ray * S * 3;
    float out[3];
    render::composite_ray(lane, S, [&](int s) { return sg[s]; }, [&](int s) { return dl[s]; },
                          [&](int s, int c) { return cl[3 * s + c]; }, weights + ray * S, out);
    if (lane == 0) {
        rgb[3 * ray + 0] = out[0];
        rgb[3 * ray + 1] = out[1];
        rgb[3 * ray + 2] = out[2];
    }
}

// Reverse of the quadrature rule.  With G_i = g_rgb . c_i (+ g_w_i):
//   dL/dc_i     = w_i g_rgb
//   dL/dsigma_i = delta_i (T_{i+1} G_i - sum_{k>i} w_k G_k),   T_{i+1} = exp(-sum_{j<=i} tau_j)
// Two passes over the ray: a forward prefix pass to get the optical depth in front of every
// 64-sample step, then steps are visited last-to-first with a suffix carry.
__global__ __launch_bounds__(RAYS_PER_BLOCK *WAVE) void composite_bwd_kernel(
    const float *__restrict__ sigma, const float *__restrict__ radiance,
    const float *__restrict__ delta, const float *__restrict__ g_rgb,
    const float *__restrict__ g_w, int64_t n, int S, float *__restrict__ g_sigma,
    float *__restrict__ g_radiance) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n) return;
    const float *sg = sigma + ray * S;
    const float *dl = delta + ray * S;
    const float *cl = radiance + ray * S * 3;
    const float g0 = g_rgb[3 * ray], g1 = g_rgb[3 * ray + 1], g2 = g_rgb[3 * ray + 2];
    const int nsteps = (S + WAVE - 1) / WAVE;
    // optical depth in front of every 64-sample step, accumulated FROM THE FRONT exactly like the forward
    // kernel's carry, and parked in lane `st` of a register.  (Rebuilding it as total - later steps cancels
    // catastrophically: the last sample has tau = sigma * 1e8, so the difference loses ~1e-6 .. 1e-4 absolute
    // and forward and backward transmittances of the 192-sample fine pass disagree.)
    double front = 0.0, before_of_lane = 0.0;
    for (int st = 0; st < nsteps; ++st) {
        const int s = st * WAVE + lane;
        const float tau = (s < S) ? sg[s] * dl[s] : 0.0f;
        if (lane == (st & (WAVE - 1)) && st < WAVE) before_of_lane = front;
        front += __shfl(wave_inclusive_scan((double)tau, lane), WAVE - 1, WAVE);
    }
    double suffix = 0.0;       // sum_{k in later steps} w_k G_k
    for (int st = nsteps - 1; st >= 0; --st) {
        const int s = st * WAVE + lane;
        const bool live = s < S;
        const float tau = live ? sg[s] * dl[s] : 0.0f;
        const double incl = wave_inclusive_scan((double)tau, lane);
        double before;           // optical depth in front of this step
        if (st < WAVE) {
            before = __shfl(before_of_lane, st, WAVE);
        } else {                 // S > 4096: walk again from the front (same additions, same order)
            before = 0.0;
            for (int e = 0; e < st; ++e) {
                const int se = e * WAVE + lane;   // e < st <= nsteps - 1: always a full step
                before += __shfl(wave_inclusive_scan((double)(sg[se] * dl[se]), lane), WAVE - 1, WAVE);
            }
        }
        double excl = __shfl_up(incl, 1, WAVE);
        if (lane == 0) excl = 0.0;
        const float T = expf(-(float)(before + excl));
        const float Tn = expf(-(float)(before + incl));  // T_{i+1}
        const float w = T * (1.0f - expf(-tau));
        float G = 0.0f;
        if (live) {
            G = g0 * cl[3 * s] + g1 * cl[3 * s + 1] + g2 * cl[3 * s + 2];
            if (g_w) G += g_w[ray * S + s];
        }
        const double wG = live ? (double)w * (double)G : 0.0;
        const double rev = wave_inclusive_scan_rev(wG, lane);   // sum_{k>=lane} in this step
        const double after = suffix + (rev - wG);                // sum_{k>i} over the whole ray
        if (live) {
            g_sigma[ray * S + s] = (float)((double)dl[s] * ((double)Tn * (double)G - after));
            g_radiance[(ray * S + s) * 3 + 0] = w * g0;
            g_radiance[(ray * S + s) * 3 + 1] = w * g1;
            g_radiance[(ray * S + s) * 3 + 2] = w * g2;
        }
        suffix += __shfl(rev, 0, WAVE);
    }
}

}  // namespace

NERF_API int nerf_composite_forward(const float *sigma, const float *radiance, const float *delta,
                                    int64_t n, int S, float *rgb, float *weights,
                                    nerf_stream_t stream) {
    NERF_REQUIRE(n >= 0 && S > 0, "nerf_composite_forward: bad sizes");
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(sigma && radiance && delta && rgb && weights, "nerf_composite_forward: null pointer");
    const unsigned grid = (unsigned)((n + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(grid), dim3(RAYS_PER_BLOCK * WAVE), 0,
                       nerf::as_stream(stream), sigma, radiance, delta, n, S, rgb, weights);
    return nerf::check_launch("nerf_composite_forward");
}

NERF_API int nerf_composite_backward(const float *sigma, const float *radiance, const float *delta,
                                     const float *g_rgb, const float *g_weights, int64_t n, int S,
                                     float *g_sigma, float *g_radiance, nerf_stream_t stream) {
    NERF_REQUIRE(n >= 0 && S > 0, "nerf_composite_backward: bad sizes");
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(sigma && radiance && delta && g_rgb && g_sigma && g_radiance,
                 "nerf_composite_backward: null pointer");
    const unsigned grid = (unsigned)((n + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(grid), dim3(RAYS_PER_BLOCK * WAVE), 0,
                       nerf::as_stream(stream), sigma, radiance, delta, g_rgb, g_weights, n, S, g_sigma,
                       g_radiance);
    return nerf::check_launch("nerf_composite_backward");
}
