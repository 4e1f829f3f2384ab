// a6 / a7: stratified and hierarchical (inverse-CDF) sampling along rays.
//
// One wavefront per ray.  The ray's sample row lives in LDS; everything that
// decides an integer in the reference (the pdf normaliser, the cdf, the bin search)
// follows ATen's CPU evaluation order literally, with separately rounded fp32
// operations, so that bin indices are bit-identical to the reference's CPU path.
// HBM traffic per ray: u1/u2/u3 in, pts+dirs+delta out (28 B/sample) -- bandwidth
// bound, tiny next to the MLP.
#include "common.h"
#include "render_device.h"

namespace {

using render::wave_fence;

// stratified_sampler.py:112-126 -- delta, sample points, repeated directions
__device__ __forceinline__ void write_samples(const float *t, int S, int lane, int64_t ray,
                                              const float *ray_o, const float *ray_d, float *t_out,
                                              float *pts, float *dirs, float *delta) {
    const float o0 = ray_o[3 * ray], o1 = ray_o[3 * ray + 1], o2 = ray_o[3 * ray + 2];
    const float d0 = ray_d[3 * ray], d1 = ray_d[3 * ray + 1], d2 = ray_d[3 * ray + 2];
    for (int s = lane; s < S; s += WAVE) {
        const float nxt = (s + 1 < S) ? t[s + 1] : 1e8f;
        delta[ray * S + s] = __fsub_rn(nxt, t[s]);
        if (t_out) t_out[ray * S + s] = t[s];
    }
    const int64_t base = ray * S * 3;
    for (int e = lane; e < 3 * S; e += WAVE) {
        const int s = e / 3, c = e - 3 * s;
        const float oc = c == 0 ? o0 : (c == 1 ? o1 : o2);
        const float dc = c == 0 ? d0 : (c == 1 ? d1 : d2);
        pts[base + e] = __fadd_rn(oc, __fmul_rn(t[s], dc));
        dirs[base + e] = dc;
    }
}

__global__ __launch_bounds__(WAVE) void stratified_kernel(const float *ray_o, const float *ray_d,
                                                          int64_t n, int S, const float *t_bins,
                                                          float ps, const float *u1, float *t_out,
                                                          float *pts, float *dirs, float *delta) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x;
    for (int64_t ray = blockIdx.x; ray < n; ray += gridDim.x) {
        wave_fence();   // the previous ray's row has been written out
        render::stratified_ray(lane, S, t_bins, ps, u1 + ray * S, sm);  // :109
        write_samples(sm, S, lane, ray, ray_o, ray_d, t_out, pts, dirs, delta);
    }
}

constexpr int HIER_RAYS_PER_BLOCK = 4;   // one wavefront per ray, four rays per workgroup (like the integral kernels)

__global__ __launch_bounds__(HIER_RAYS_PER_BLOCK * WAVE) void hierarchical_kernel(
    const float *ray_o, const float *ray_d, int64_t n, int Sc, int Sf, const float *t_bins, float ps,
    float *weights, const float *u1, const float *u2, const float *u3, int64_t *bin_idx, float *t_out,
    float *pts, float *dirs, float *delta, int floats_per_ray) {
    extern __shared__ __attribute__((aligned(16))) float sm_all[];
    const int S = Sc + Sf;
    const int wave = threadIdx.x / WAVE, rays_per_block = blockDim.x / WAVE;
    float *sm = sm_all + wave * floats_per_ray;
    float *t_srt = sm;           // S sorted positions; the per-ray scratch rows follow
    float *scratch = sm + ((S + 3) & ~3);
    const int lane = threadIdx.x & (WAVE - 1);
    for (int64_t ray = (int64_t)blockIdx.x * rays_per_block + wave; ray < n; ray += (int64_t)gridDim.x * rays_per_block) {
        wave_fence();   // the previous ray's row has been written out
        render::hierarchical_ray(lane, Sc, Sf, t_bins, ps, weights + ray * Sc, u1 + ray * Sc, u2 + ray * Sf,
                                 u3 + ray * Sf, bin_idx ? bin_idx + ray * Sf : nullptr, scratch, t_srt);
        write_samples(t_srt, S, lane, ray, ray_o, ray_d, t_out, pts, dirs, delta);
    }
}

constexpr int MAX_LDS_FLOATS = 16000;  // stay under the 64 KiB default dynamic-LDS limit

}  // namespace

NERF_API int nerf_sample_stratified(const float *ray_o, const float *ray_d, int64_t n, int S,
                                    const float *t_bins, float partition_size, const float *u1,
                                    float *t, float *pts, float *dirs, float *delta,
                                    nerf_stream_t stream) {
    NERF_REQUIRE(n >= 0 && S > 0, "nerf_sample_stratified: bad sizes");
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(ray_o && ray_d && t_bins && u1 && pts && dirs && delta,
                 "nerf_sample_stratified: null pointer");
    if (S > MAX_LDS_FLOATS) return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_sample_stratified: S too large");
    const unsigned grid = (unsigned)(n < 1048576 ? n : 1048576);
    hipLaunchKernelGGL(stratified_kernel, dim3(grid), dim3(WAVE), (size_t)S * sizeof(float),
                       nerf::as_stream(stream), ray_o, ray_d, n, S, t_bins, partition_size, u1, t, pts,
                       dirs, delta);
    return nerf::check_launch("nerf_sample_stratified");
}

NERF_API int nerf_sample_hierarchical(const float *ray_o, const float *ray_d, int64_t n, int Sc, int Sf,
                                      const float *t_bins, float partition_size, float *weights,
                                      const float *u1, const float *u2, const float *u3,
                                      int64_t *bin_idx, float *t, float *pts, float *dirs, float *delta,
                                      nerf_stream_t stream) {
    NERF_REQUIRE(n >= 0 && Sc > 0 && Sf >= 0, "nerf_sample_hierarchical: bad sizes");
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(ray_o && ray_d && t_bins && weights && u1 && (Sf == 0 || (u2 && u3)) && pts && dirs &&
                     delta,
                 "nerf_sample_hierarchical: null pointer");
    const size_t floats = (size_t)((Sc + Sf + 3) & ~3) + (size_t)render::hierarchical_scratch_floats(Sc, Sf);
    if (floats > (size_t)MAX_LDS_FLOATS)
        return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_sample_hierarchical: Sc+Sf too large for LDS");
    const int rpb = HIER_RAYS_PER_BLOCK * floats <= (size_t)MAX_LDS_FLOATS ? HIER_RAYS_PER_BLOCK : 1;   // very long rows: one ray per workgroup
    const int64_t blocks = (n + rpb - 1) / rpb;
    const unsigned grid = (unsigned)(blocks < 1048576 ? blocks : 1048576);
    hipLaunchKernelGGL(hierarchical_kernel, dim3(grid), dim3(rpb * WAVE),
                       rpb * floats * sizeof(float), nerf::as_stream(stream), ray_o, ray_d, n, Sc, Sf,
                       t_bins, partition_size, weights, u1, u2, u3, bin_idx, t, pts, dirs, delta, (int)floats);
    return nerf::check_launch("nerf_sample_hierarchical");
}
