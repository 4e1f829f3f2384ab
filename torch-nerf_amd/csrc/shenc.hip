// Spherical-harmonics encoder (SHEncoder.encode, R/signal_encoder/spherical_harmonics_encoder.py:86-139): the
// encoder the runners build for BOTH inputs under `signal_encoder: sh` (R/../runners/runner_utils.py:595-604,
// configs/signal_encoder/sh.yaml: degree 4 -> 16 features), forward and the gradient autograd returns for in_signal.
// One thread per sample: 12 B in, 4 degree^2 B out, HBM-bound like posenc.hip.  The products are evaluated in the
// reference's order (python scalar times tensor first, then the polynomial factor; the library is built with
// -ffp-contract=off), so the values are bit-identical to the reference's fp32 result.
#include "common.h"

namespace {

// real SH basis constants, l = 0 .. 4 (the reference's coeff_0 .. coeff_4 tables: standard values, e.g. 0.5 sqrt(1/pi))
__device__ __constant__ const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
__device__ __constant__ const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                             -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                             0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                             -0.5900435899266435f};
__device__ __constant__ const float C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f,
                                             -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f,
                                             0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};

__global__ void shenc_kernel(const float *__restrict__ in, int64_t M, int degree, float *__restrict__ out) {
    const int E = degree * degree;
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
        const float x = in[3 * m], y = in[3 * m + 1], z = in[3 * m + 2];
        float *e = out + m * E;
        e[0] = C0;                                                            // :106
        if (degree <= 1) continue;
        e[1] = -C1 * y; e[2] = C1 * z; e[3] = -C1 * x;                        // :108-110
        if (degree <= 2) continue;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;   // :112-113
        e[4] = C2[0] * xy;                                                    // :114-118
        e[5] = C2[1] * yz;
        e[6] = C2[2] * ((2.0f * zz - xx) - yy);
        e[7] = C2[3] * xz;
        e[8] = C2[4] * (xx - yy);
        if (degree <= 3) continue;
        e[9] = (C3[0] * y) * (3.0f * xx - yy);                                // :120-126
        e[10] = (C3[1] * xy) * z;
        e[11] = (C3[2] * y) * ((4.0f * zz - xx) - yy);
        e[12] = (C3[3] * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
        e[13] = (C3[4] * x) * ((4.0f * zz - xx) - yy);
        e[14] = (C3[5] * z) * (xx - yy);
        e[15] = (C3[6] * x) * (xx - 3.0f * yy);
        if (degree <= 4) continue;
        e[16] = (C4[0] * xy) * (xx - yy);                                     // :128-138
        e[17] = (C4[1] * yz) * (3.0f * xx - yy);
        e[18] = (C4[2] * xy) * (7.0f * zz - 1.0f);
        e[19] = (C4[3] * yz) * (7.0f * zz - 3.0f);
        e[20] = C4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f);
        e[21] = (C4[5] * xz) * (7.0f * zz - 3.0f);
        e[22] = (C4[6] * (xx - yy)) * (7.0f * zz - 1.0f);
        e[23] = (C4[7] * xz) * (xx - 3.0f * yy);
        e[24] = C4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
    }
}

// g_in[m] = sum_k g_out[m][k] * d e_k / d (x, y, z): analytic derivatives of the polynomials above
__global__ void shenc_bwd_kernel(const float *__restrict__ in, const float *__restrict__ g_out, int64_t M, int degree,
                                 float *__restrict__ g_in) {
    const int E = degree * degree;
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
        const float x = in[3 * m], y = in[3 * m + 1], z = in[3 * m + 2];
        const float *g = g_out + m * E;
        float gx = 0.0f, gy = 0.0f, gz = 0.0f;
        if (degree > 1) {
            gy += g[1] * -C1; gz += g[2] * C1; gx += g[3] * -C1;
        }
        if (degree > 2) {
            gx += g[4] * C2[0] * y;            gy += g[4] * C2[0] * x;
            gy += g[5] * C2[1] * z;            gz += g[5] * C2[1] * y;
            gx += g[6] * C2[2] * (-2.0f * x);  gy += g[6] * C2[2] * (-2.0f * y);  gz += g[6] * C2[2] * (4.0f * z);
            gx += g[7] * C2[3] * z;            gz += g[7] * C2[3] * x;
            gx += g[8] * C2[4] * (2.0f * x);   gy += g[8] * C2[4] * (-2.0f * y);
        }
        if (degree > 3) {
            const float xx = x * x, yy = y * y, zz = z * z;
            gx += g[9] * C3[0] * (6.0f * x * y);              gy += g[9] * C3[0] * (3.0f * xx - 3.0f * yy);
            gx += g[10] * C3[1] * (y * z);  gy += g[10] * C3[1] * (x * z);  gz += g[10] * C3[1] * (x * y);
            gx += g[11] * C3[2] * (-2.0f * x * y);  gy += g[11] * C3[2] * (4.0f * zz - xx - 3.0f * yy);  gz += g[11] * C3[2] * (8.0f * y * z);
            gx += g[12] * C3[3] * (-6.0f * x * z);  gy += g[12] * C3[3] * (-6.0f * y * z);  gz += g[12] * C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
            gx += g[13] * C3[4] * (4.0f * zz - 3.0f * xx - yy);  gy += g[13] * C3[4] * (-2.0f * x * y);  gz += g[13] * C3[4] * (8.0f * x * z);
            gx += g[14] * C3[5] * (2.0f * x * z);  gy += g[14] * C3[5] * (-2.0f * y * z);  gz += g[14] * C3[5] * (xx - yy);
            gx += g[15] * C3[6] * (3.0f * xx - 3.0f * yy);        gy += g[15] * C3[6] * (-6.0f * x * y);
        }
        if (degree > 4) {
            const float xx = x * x, yy = y * y, zz = z * z;
            gx += g[16] * C4[0] * (3.0f * xx * y - yy * y);   gy += g[16] * C4[0] * (xx * x - 3.0f * x * yy);
            gx += g[17] * C4[1] * (6.0f * x * y * z);  gy += g[17] * C4[1] * (3.0f * xx * z - 3.0f * yy * z);  gz += g[17] * C4[1] * (3.0f * xx * y - yy * y);
            gx += g[18] * C4[2] * (y * (7.0f * zz - 1.0f));  gy += g[18] * C4[2] * (x * (7.0f * zz - 1.0f));  gz += g[18] * C4[2] * (14.0f * x * y * z);
            gy += g[19] * C4[3] * (z * (7.0f * zz - 3.0f));  gz += g[19] * C4[3] * (y * (21.0f * zz - 3.0f));
            gz += g[20] * C4[4] * (140.0f * zz * z - 60.0f * z);
            gx += g[21] * C4[5] * (z * (7.0f * zz - 3.0f));  gz += g[21] * C4[5] * (x * (21.0f * zz - 3.0f));
            gx += g[22] * C4[6] * (2.0f * x * (7.0f * zz - 1.0f));  gy += g[22] * C4[6] * (-2.0f * y * (7.0f * zz - 1.0f));  gz += g[22] * C4[6] * (14.0f * z * (xx - yy));
            gx += g[23] * C4[7] * (3.0f * xx * z - 3.0f * yy * z);  gy += g[23] * C4[7] * (-6.0f * x * y * z);  gz += g[23] * C4[7] * (xx * x - 3.0f * x * yy);
            gx += g[24] * C4[8] * (4.0f * xx * x - 12.0f * x * yy);  gy += g[24] * C4[8] * (4.0f * yy * y - 12.0f * xx * y);
        }
        g_in[3 * m] = gx; g_in[3 * m + 1] = gy; g_in[3 * m + 2] = gz;
    }
}

}  // namespace

NERF_API int nerf_shenc(const float *in_signal, int64_t M, int degree, float *out, nerf_stream_t stream) {
    NERF_REQUIRE(M >= 0 && degree >= 1 && degree <= 5, "nerf_shenc: degree 1..5 (the reference defines no others)");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(in_signal && out, "nerf_shenc: null pointer");
    int64_t grid = (M + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(shenc_kernel, dim3((unsigned)grid), dim3(256), 0, nerf::as_stream(stream), in_signal, M, degree, out);
    return nerf::check_launch("nerf_shenc");
}

NERF_API int nerf_shenc_backward(const float *in_signal, const float *g_out, int64_t M, int degree, float *g_in,
                                 nerf_stream_t stream) {
    NERF_REQUIRE(M >= 0 && degree >= 1 && degree <= 5, "nerf_shenc_backward: degree 1..5");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(in_signal && g_out && g_in, "nerf_shenc_backward: null pointer");
    int64_t grid = (M + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(shenc_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, nerf::as_stream(stream), in_signal, g_out, M,
                       degree, g_in);
    return nerf::check_launch("nerf_shenc_backward");
}
