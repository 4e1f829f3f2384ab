// a2 / a4 / a5: screen coordinates, ray generation, NDC projection.
// One thread per ray; 24 B written per ray -- launch-latency bound, not bandwidth bound.
#include "common.h"

namespace {

struct Camera {
    float fx, fy, cx, cy;
    float e[12];  // row-major [R|t]
    float ndc_sx, ndc_sy, ndc_tn;
    int ndc;
};

__device__ __forceinline__ void pixel_to_coord(const int64_t *coords, const int64_t *pix,
                                               int64_t first, int64_t i, int64_t H, int64_t W,
                                               int64_t &u, int64_t &v) {
    if (coords) {
        u = coords[2 * i];
        v = coords[2 * i + 1];
    } else {
        const int64_t p = pix ? pix[i] : first + i;
        u = p % W;
        v = (H - 1) - p / W;  // volume_renderer.py:183 flips rows
    }
}

__global__ void screen_coords_kernel(int64_t H, int64_t W, const int64_t *pix, int64_t first,
                                     int64_t n, int64_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t u, v;
    pixel_to_coord(nullptr, pix, first, i, H, W, u, v);
    out[2 * i] = u;
    out[2 * i + 1] = v;
}

__global__ void raygen_kernel(const int64_t *coords, const int64_t *pix, int64_t first, int64_t n,
                              int64_t H, int64_t W, Camera cam, float *ro, float *rd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t u, v;
    pixel_to_coord(coords, pix, first, i, H, W, u, v);
    // sampler_base.py:92-94: (u - cx)/fx, (v - cy)/fy, -1 ; no pixel-centre offset, no normalise
    const float x = __fdiv_rn(__fsub_rn((float)u, cam.cx), cam.fx);
    const float y = __fdiv_rn(__fsub_rn((float)v, cam.cy), cam.fy);
    float o[3], d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        // sampler_base.py:164-165: d = d_cam @ R^T ; o = 0 + t   (separately rounded)
        float acc = __fmul_rn(x, cam.e[4 * r + 0]);
        acc = __fadd_rn(acc, __fmul_rn(y, cam.e[4 * r + 1]));
        acc = __fadd_rn(acc, __fmul_rn(-1.0f, cam.e[4 * r + 2]));
        d[r] = acc;
        o[r] = cam.e[4 * r + 3];
    }
    if (cam.ndc) {
        // sampler_base.py:236-255
        const float oxz = __fdiv_rn(o[0], o[2]), oyz = __fdiv_rn(o[1], o[2]);
        const float tz = __fdiv_rn(cam.ndc_tn, o[2]);
        const float nd0 = __fmul_rn(cam.ndc_sx, __fsub_rn(__fdiv_rn(d[0], d[2]), oxz));
        const float nd1 = __fmul_rn(cam.ndc_sy, __fsub_rn(__fdiv_rn(d[1], d[2]), oyz));
        o[0] = __fmul_rn(cam.ndc_sx, oxz);
        o[1] = __fmul_rn(cam.ndc_sy, oyz);
        o[2] = __fadd_rn(1.0f, tz);
        d[0] = nd0;
        d[1] = nd1;
        d[2] = -tz;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        ro[3 * i + r] = o[r];
        rd[3 * i + r] = d[r];
    }
}

}  // namespace

NERF_API int nerf_screen_coords(int64_t H, int64_t W, const int64_t *pix, int64_t first, int64_t n,
                                int64_t *coords, nerf_stream_t stream) {
    NERF_REQUIRE(H > 0 && W > 0 && n >= 0 && (coords || n == 0), "nerf_screen_coords: bad arguments");
    if (n == 0) return NERF_OK;
    const int block = 256;
    hipLaunchKernelGGL(screen_coords_kernel, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0,
                       nerf::as_stream(stream), H, W, pix, first, n, coords);
    return nerf::check_launch("nerf_screen_coords");
}

NERF_API int nerf_generate_rays(const int64_t *coords, const int64_t *pix, int64_t first, int64_t n,
                                int64_t H, int64_t W, float fx, float fy, float cx, float cy,
                                const float *extrinsic_host, int project_to_ndc, double focal,
                                double z_near, float *ray_o, float *ray_d, nerf_stream_t stream) {
    NERF_REQUIRE(n >= 0 && H > 0 && W > 0 && extrinsic_host, "nerf_generate_rays: bad arguments");
    NERF_REQUIRE(n == 0 || (ray_o && ray_d), "nerf_generate_rays: null output");
    if (project_to_ndc && z_near < 0.0)  // sampler_base.py:232-233
        return nerf::fail(NERF_ERR_ARG, "nerf_generate_rays: z_near must be >= 0");
    if (n == 0) return NERF_OK;
    Camera cam;
    cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy;
    for (int k = 0; k < 12; ++k) cam.e[k] = extrinsic_host[k];
    cam.ndc = project_to_ndc ? 1 : 0;
    // python floats are rounded to fp32 when they meet a tensor
    cam.ndc_sx = (float)(-(2.0 * focal / (double)W));
    cam.ndc_sy = (float)(-(2.0 * focal / (double)H));
    cam.ndc_tn = (float)(2.0 * z_near);
    const int block = 256;
    hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0,
                       nerf::as_stream(stream), coords, pix, first, n, H, W, cam, ray_o, ray_d);
    return nerf::check_launch("nerf_generate_rays");
}
