// a6 / a7: stratified and hierarchical (inverse-CDF) sampling along rays.
//
// One wavefront per ray.  The ray's sample row lives in LDS; everything that
// decides an integer in the reference (the pdf normaliser, the cdf, the bin search)
// follows ATen's CPU evaluation order literally, with separately rounded fp32
// operations, so that bin indices are bit-identical to the reference's CPU path.
// HBM traffic per ray: u1/u2/u3 in, pts+dirs+delta out (28 B/sample) -- bandwidth
// bound, tiny next to the MLP.
#include "common.h"

namespace {

__device__ __forceinline__ int ceil_log2_i(int x) {
    if (x <= 2) return 1;
    return 32 - __builtin_clz((unsigned)(x - 1));
}

// ATen's CPU sum over a contiguous last dimension (SumKernel.cpp: vectorized_inner_sum ->
// row_sum -> multi_row_sum, 8-wide vectors, 4 interleaved accumulators, 4-level cascade).
// Lane l < 8 plays vector lane l and returns its partial; the caller finishes on lane 0.
__device__ float aten_sum_vector_lane(const float *row, int size0, int l) {
    constexpr int VEC = 8, ILP = 4, LEVELS = 4;
    const int vec_size = size0 / VEC;
    const int size_ilp = vec_size / ILP;
    int level_power = ceil_log2_i(size_ilp) / LEVELS;
    if (level_power < 4) level_power = 4;
    const int level_step = 1 << level_power;
    const int level_mask = level_step - 1;
    float acc[LEVELS][ILP];
#pragma unroll
    for (int j = 0; j < LEVELS; ++j)
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc[j][k] = 0.0f;
    int i = 0;
    for (; i + level_step <= size_ilp;) {
        for (int j = 0; j < level_step; ++j, ++i) {
#pragma unroll
            for (int k = 0; k < ILP; ++k)
                acc[0][k] = __fadd_rn(acc[0][k], row[(i * ILP + k) * VEC + l]);
        }
        bool stop = false;
#pragma unroll
        for (int j = 1; j < LEVELS; ++j) {
            if (!stop) {
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    acc[j][k] = __fadd_rn(acc[j][k], acc[j - 1][k]);
                    acc[j - 1][k] = 0.0f;
                }
                const int mask = level_mask << (j * level_power);
                if ((i & mask) != 0) stop = true;
            }
        }
    }
    for (; i < size_ilp; ++i) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc[0][k] = __fadd_rn(acc[0][k], row[(i * ILP + k) * VEC + l]);
    }
#pragma unroll
    for (int j = 1; j < LEVELS; ++j)
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc[0][k] = __fadd_rn(acc[0][k], acc[j][k]);
    // row_sum tail: whole vectors left over after the (-1, ILP) view
    for (int v = size_ilp * ILP; v < vec_size; ++v) acc[0][0] = __fadd_rn(acc[0][0], row[v * VEC + l]);
#pragma unroll
    for (int k = 1; k < ILP; ++k) acc[0][0] = __fadd_rn(acc[0][0], acc[0][k]);
    return acc[0][0];
}

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
        __syncthreads();
        for (int s = lane; s < S; s += WAVE)
            sm[s] = __fadd_rn(t_bins[s], __fmul_rn(ps, u1[ray * S + s]));  // :109
        __syncthreads();
        write_samples(sm, S, lane, ray, ray_o, ray_d, t_out, pts, dirs, delta);
    }
}

__global__ __launch_bounds__(WAVE) void hierarchical_kernel(
    const float *ray_o, const float *ray_d, int64_t n, int Sc, int Sf, const float *t_bins, float ps,
    float *weights, const float *u1, const float *u2, const float *u3, int64_t *bin_idx, float *t_out,
    float *pts, float *dirs, float *delta) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int S = Sc + Sf;
    float *t_raw = sm;           // S   coarse then fine, unsorted
    float *t_srt = t_raw + S;    // S   sorted
    float *w = t_srt + S;        // Sc  weights + 1e-5, then pdf
    float *cdf = w + Sc;         // Sc
    float *part = cdf + Sc;      // 8 partial sums + 1 normaliser
    const int lane = threadIdx.x;
    for (int64_t ray = blockIdx.x; ray < n; ray += gridDim.x) {
        __syncthreads();
        // utils.py:31  weights += 1e-5 (in place, visible to the caller)
        for (int s = lane; s < Sc; s += WAVE) {
            const float v = __fadd_rn(weights[ray * Sc + s], 1e-5f);
            weights[ray * Sc + s] = v;
            w[s] = v;
            // stratified_sampler.py:77  new coarse jitter
            t_raw[s] = __fadd_rn(t_bins[s], __fmul_rn(ps, u1[ray * Sc + s]));
        }
        __syncthreads();
        // utils.py:32  normalizer = torch.sum(weights, -1) in ATen's order
        if (lane < 8) part[lane] = aten_sum_vector_lane(w, Sc, lane);
        __syncthreads();
        if (lane == 0) {
            float fin = 0.0f;
            for (int k = (Sc / 8) * 8; k < Sc; ++k) fin = __fadd_rn(fin, w[k]);
            for (int l = 0; l < 8; ++l) fin = __fadd_rn(fin, part[l]);
            part[8] = fin;
        }
        __syncthreads();
        const float norm = part[8];
        for (int s = lane; s < Sc; s += WAVE) w[s] = __fdiv_rn(w[s], norm);  // utils.py:33
        __syncthreads();
        // utils.py:36-40  cdf = [0, cumsum(pdf)[:-1]] ; ATen CPU cumsum: double accumulator,
        // every prefix rounded to fp32
        if (lane == 0) {
            double run = 0.0;
            cdf[0] = 0.0f;
            for (int s = 0; s + 1 < Sc; ++s) {
                run += (double)w[s];
                cdf[s + 1] = (float)run;
            }
        }
        __syncthreads();
        // utils.py:43-56  searchsorted(right=True) - 1, gather, in-bin jitter
        for (int f = lane; f < Sf; f += WAVE) {
            const float y = u2[ray * Sf + f];
            int lo = 0, hi = Sc;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] <= y) lo = mid + 1; else hi = mid;
            }
            int k = lo - 1;
            if (bin_idx) bin_idx[ray * Sf + f] = (int64_t)k;
            if (k < 0) k = 0;
            t_raw[Sc + f] = __fadd_rn(t_bins[k], __fmul_rn(ps, u3[ray * Sf + f]));
        }
        __syncthreads();
        // stratified_sampler.py:87-90  sort(cat[coarse, fine]) -- rank sort, S^2/64 compares
        for (int e = lane; e < S; e += WAVE) {
            const float x = t_raw[e];
            int rank = 0;
            for (int j = 0; j < S; ++j) {
                const float xj = t_raw[j];
                rank += (xj < x || (xj == x && j < e)) ? 1 : 0;
            }
            t_srt[rank] = x;
        }
        __syncthreads();
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
    const size_t floats = (size_t)2 * (Sc + Sf) + 2 * (size_t)Sc + 16;
    if (floats > (size_t)MAX_LDS_FLOATS)
        return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_sample_hierarchical: Sc+Sf too large for LDS");
    const unsigned grid = (unsigned)(n < 1048576 ? n : 1048576);
    hipLaunchKernelGGL(hierarchical_kernel, dim3(grid), dim3(WAVE), floats * sizeof(float),
                       nerf::as_stream(stream), ray_o, ray_d, n, Sc, Sf, t_bins, partition_size, weights,
                       u1, u2, u3, bin_idx, t, pts, dirs, delta);
    return nerf::check_launch("nerf_sample_hierarchical");
}
