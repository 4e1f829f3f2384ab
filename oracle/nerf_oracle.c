/*
 * nerf_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, CPU restatement of the NeRF volume-rendering hot path of
 * DveloperY0115/torch-NeRF (reference tree: torch_nerf/src/...).  It exists to
 * CHECK the HIP kernels; it is never the thing shipped or measured.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED against outputs of the reference itself, imported in
 * the build container (tests/golden/make_golden.py -> tests/golden/ npz files, see
 * tests/test_oracle_golden.py).  The reference ships no tests or golden vectors
 * of its own, and its arithmetic lives in PyTorch/ATen
 * (third party, torch 2.10.0 CPU build here; the reference pins torch 1.11.0).
 * Where ATen's CPU summation order decides an integer result (fine-sample bin
 * indices) the order is restated explicitly below and cited.
 *
 * All arithmetic is IEEE fp32 with separate multiply/add roundings
 * (build with -ffp-contract=off) unless a comment says otherwise.
 *
 * Every function cites the reference file:line it follows
 * (R/ = torch_nerf/src/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(_OPENMP)
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* a2: screen coordinates.  R/renderer/volume_renderer.py:171-190      */
/* coords[i] = (i % W, (H-1) - i / W)                                  */
/* ------------------------------------------------------------------ */
ORC_API void orc_screen_coords(const int64_t *pix, int64_t n, int64_t H, int64_t W,
                               int64_t *coords /* (n,2) */)
{
    for (int64_t i = 0; i < n; ++i) {
        int64_t p = pix ? pix[i] : i;
        coords[2 * i + 0] = p % W;
        coords[2 * i + 1] = (H - 1) - p / W;
    }
}

/* ------------------------------------------------------------------ */
/* a4: ray generation.  R/renderer/ray_samplers/sampler_base.py:70-113 */
/* (directions), :134-166 (camera -> world).  No normalisation, no     */
/* +0.5 pixel centre.  extr is the row-major 3x4 [R|t] block.          */
/* d = d_cam @ R^T is evaluated as a left-to-right fp32 dot product;   */
/* ATen's sgemm may fuse/reorder it, so rays are a tolerance quantity. */
/* ------------------------------------------------------------------ */
ORC_API void orc_raygen(const int64_t *coords, int64_t n, float fx, float fy, float cx, float cy,
                        const float *extr /* 3x4 */, float *o, float *d)
{
    for (int64_t i = 0; i < n; ++i) {
        float x = ((float)coords[2 * i + 0] - cx) / fx;
        float y = ((float)coords[2 * i + 1] - cy) / fy;
        float z = -1.0f;
        for (int r = 0; r < 3; ++r) {
            float acc = x * extr[4 * r + 0];
            acc = acc + y * extr[4 * r + 1];
            acc = acc + z * extr[4 * r + 2];
            d[3 * i + r] = acc;
            o[3 * i + r] = 0.0f + extr[4 * r + 3];
        }
    }
}

/* ------------------------------------------------------------------ */
/* a5: NDC projection.  R/renderer/ray_samplers/sampler_base.py:199-257*/
/* Python-float scalars are rounded to fp32 before meeting a tensor.   */
/* In place on (o, d).                                                 */
/* ------------------------------------------------------------------ */
ORC_API void orc_map_rays_to_ndc(double focal, double z_near, int64_t H, int64_t W, int64_t n,
                                 float *o, float *d)
{
    const float sx = (float)(-(2.0 * focal / (double)W));
    const float sy = (float)(-(2.0 * focal / (double)H));
    const float tn = (float)(2.0 * z_near);
    for (int64_t i = 0; i < n; ++i) {
        float ox = o[3 * i], oy = o[3 * i + 1], oz = o[3 * i + 2];
        float dx = d[3 * i], dy = d[3 * i + 1], dz = d[3 * i + 2];
        float oxz = ox / oz, oyz = oy / oz;
        float no_x = sx * oxz;
        float no_y = sy * oyz;
        float no_z = 1.0f + (tn / oz);
        float nd_x = sx * ((dx / dz) - oxz);
        float nd_y = sy * ((dy / dz) - oyz);
        float nd_z = -(tn / oz);
        o[3 * i] = no_x; o[3 * i + 1] = no_y; o[3 * i + 2] = no_z;
        d[3 * i] = nd_x; d[3 * i + 1] = nd_y; d[3 * i + 2] = nd_z;
    }
}

/* shared epilogue of both sampling branches:                          */
/* R/renderer/ray_samplers/stratified_sampler.py:112-126               */
static void finish_samples(const float *o, const float *d, const float *t, int64_t S,
                           float *pts, float *dirs, float *delta)
{
    for (int64_t s = 0; s < S; ++s) {
        float nxt = (s + 1 < S) ? t[s + 1] : 1e8f;
        delta[s] = nxt - t[s];
        for (int c = 0; c < 3; ++c) {
            float td = t[s] * d[c];
            pts[3 * s + c] = o[c] + td;
            dirs[3 * s + c] = d[c];
        }
    }
}

/* ------------------------------------------------------------------ */
/* a6: stratified (coarse) sampling.                                   */
/* R/renderer/ray_samplers/stratified_sampler.py:91-128, :130-164      */
/* t_bins (S,) comes from torch.linspace on the host side of the       */
/* boundary; ps = fp32(partition_size); u1 = rand_like(t_bins) (N,S).  */
/* ------------------------------------------------------------------ */
ORC_API void orc_stratified_sample(const float *o, const float *d, int64_t n, int64_t S,
                                   const float *t_bins, float ps, const float *u1,
                                   float *t_out, float *pts, float *dirs, float *delta)
{
    for (int64_t i = 0; i < n; ++i) {
        float *t = t_out + i * S;
        for (int64_t s = 0; s < S; ++s) {
            float j = ps * u1[i * S + s];
            t[s] = t_bins[s] + j;
        }
        finish_samples(o + 3 * i, d + 3 * i, t, S, pts + 3 * i * S, dirs + 3 * i * S,
                       delta + i * S);
    }
}

/* ------------------------------------------------------------------ */
/* ATen CPU float sum over a contiguous last dim, restated.            */
/* aten/src/ATen/native/cpu/SumKernel.cpp (torch 2.10, AVX2 dispatch): */
/* vectorized_inner_sum -> row_sum (ilp_factor 4) -> multi_row_sum     */
/* (cascade, 4 levels, level_power = max(4, ceil_log2(size)/4)).       */
/* Vector width 8 floats.  This order decides torch.sum(weights,-1) in */
/* R/renderer/ray_samplers/utils.py:32 and hence the fine-sample bins. */
/* ------------------------------------------------------------------ */
static int ceil_log2_i64(int64_t x)
{
    if (x <= 2) return 1;
    int l = 0;
    int64_t v = x - 1;
    while (v > 0) { v >>= 1; ++l; }
    return l;
}

#define ORC_VEC 8
#define ORC_ILP 4
#define ORC_LEVELS 4

ORC_API float orc_aten_sum_lastdim(const float *row, int64_t size0)
{
    const int64_t vec_size = size0 / ORC_VEC;       /* number of whole vectors        */
    const int64_t size_ilp = vec_size / ORC_ILP;    /* rows of the (-1, ilp) view     */
    float part[ORC_ILP][ORC_VEC];
    /* ---- multi_row_sum over `size_ilp` steps, ORC_ILP interleaved rows ---- */
    {
        int level_power = ceil_log2_i64(size_ilp) / ORC_LEVELS;
        if (level_power < 4) level_power = 4;
        const int64_t level_step = (int64_t)1 << level_power;
        const int64_t level_mask = level_step - 1;
        float acc[ORC_LEVELS][ORC_ILP][ORC_VEC];
        memset(acc, 0, sizeof(acc));
        int64_t i = 0;
        for (; i + level_step <= size_ilp;) {
            for (int64_t j = 0; j < level_step; ++j, ++i)
                for (int k = 0; k < ORC_ILP; ++k)
                    for (int l = 0; l < ORC_VEC; ++l)
                        acc[0][k][l] = acc[0][k][l] + row[(i * ORC_ILP + k) * ORC_VEC + l];
            for (int j = 1; j < ORC_LEVELS; ++j) {
                for (int k = 0; k < ORC_ILP; ++k)
                    for (int l = 0; l < ORC_VEC; ++l) {
                        acc[j][k][l] = acc[j][k][l] + acc[j - 1][k][l];
                        acc[j - 1][k][l] = 0.0f;
                    }
                const int64_t mask = level_mask << (j * level_power);
                if ((i & mask) != 0) break;
            }
        }
        for (; i < size_ilp; ++i)
            for (int k = 0; k < ORC_ILP; ++k)
                for (int l = 0; l < ORC_VEC; ++l)
                    acc[0][k][l] = acc[0][k][l] + row[(i * ORC_ILP + k) * ORC_VEC + l];
        for (int j = 1; j < ORC_LEVELS; ++j)
            for (int k = 0; k < ORC_ILP; ++k)
                for (int l = 0; l < ORC_VEC; ++l)
                    acc[0][k][l] = acc[0][k][l] + acc[j][k][l];
        memcpy(part, acc[0], sizeof(part));
    }
    /* ---- row_sum tail: leftover whole vectors go to partial 0 ---- */
    for (int64_t v = size_ilp * ORC_ILP; v < vec_size; ++v)
        for (int l = 0; l < ORC_VEC; ++l) part[0][l] = part[0][l] + row[v * ORC_VEC + l];
    for (int k = 1; k < ORC_ILP; ++k)
        for (int l = 0; l < ORC_VEC; ++l) part[0][l] = part[0][l] + part[k][l];
    /* ---- vectorized_inner_sum tail: scalars first, then the 8 lanes ---- */
    float fin = 0.0f;
    for (int64_t k = vec_size * ORC_VEC; k < size0; ++k) fin = fin + row[k];
    for (int l = 0; l < ORC_VEC; ++l) fin = fin + part[0][l];
    return fin;
}

/* ------------------------------------------------------------------ */
/* a7: hierarchical (fine) sampling.                                   */
/* R/renderer/ray_samplers/stratified_sampler.py:57-90 and             */
/* R/renderer/ray_samplers/utils.py:8-58 (sample_pdf).                 */
/*  - weights += 1e-5 IN PLACE (utils.py:31)                           */
/*  - pdf = w / sum(w)  (ATen sum order above)                         */
/*  - cdf = [0, cumsum(pdf)[:-1]]; ATen CPU cumsum accumulates in      */
/*    double and rounds each prefix to fp32                            */
/*    (aten/src/ATen/native/cpu/ReduceOpsKernel.cpp cumsum_cpu_kernel, */
/*    acc_type<float,false> = double)                                  */
/*  - idx = searchsorted(cdf, u2, right=True) - 1                      */
/*  - t_fine = t_bins[idx] + ps*u3 ; t = sort(cat[t_coarse, t_fine])   */
/* ------------------------------------------------------------------ */
static int cmp_float(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

ORC_API void orc_hierarchical_sample(const float *o, const float *d, int64_t n, int64_t Sc,
                                     int64_t Sf, const float *t_bins, float ps,
                                     float *weights /* (n,Sc) in/out */, const float *u1,
                                     const float *u2, const float *u3,
                                     int64_t *idx_out /* (n,Sf) or NULL */, float *t_out,
                                     float *pts, float *dirs, float *delta)
{
    const int64_t S = Sc + Sf;
    float *pdf = (float *)malloc(sizeof(float) * Sc);
    float *cdf = (float *)malloc(sizeof(float) * Sc);
    for (int64_t i = 0; i < n; ++i) {
        float *w = weights + i * Sc;
        float *t = t_out + i * S;
        for (int64_t s = 0; s < Sc; ++s) w[s] = w[s] + 1e-5f;
        const float norm = orc_aten_sum_lastdim(w, Sc);
        for (int64_t s = 0; s < Sc; ++s) pdf[s] = w[s] / norm;
        double run = 0.0;
        cdf[0] = 0.0f;
        for (int64_t s = 0; s + 1 < Sc; ++s) {
            run += (double)pdf[s];
            cdf[s + 1] = (float)run;
        }
        for (int64_t s = 0; s < Sc; ++s) {
            float j = ps * u1[i * Sc + s];
            t[s] = t_bins[s] + j;
        }
        for (int64_t f = 0; f < Sf; ++f) {
            const float y = u2[i * Sf + f];
            int64_t cnt = 0; /* number of cdf entries <= y  (upper bound) */
            for (int64_t s = 0; s < Sc; ++s) cnt += (cdf[s] <= y) ? 1 : 0;
            int64_t k = cnt - 1;
            if (idx_out) idx_out[i * Sf + f] = k;
            /* torch.gather would raise on k = -1 (y < 0 never happens for U[0,1)) */
            if (k < 0) k = 0;
            float j = ps * u3[i * Sf + f];
            t[Sc + f] = t_bins[k] + j;
        }
        qsort(t, (size_t)S, sizeof(float), cmp_float);
        finish_samples(o + 3 * i, d + 3 * i, t, S, pts + 3 * i * S, dirs + 3 * i * S,
                       delta + i * S);
    }
    free(pdf);
    free(cdf);
}

/* ------------------------------------------------------------------ */
/* a8: positional encoding.                                            */
/* R/signal_encoder/positional_encoder.py:49-104                       */
/* out = [x, sin(1x), cos(1x), sin(2x), cos(2x), ... ] ; each block is */
/* all C channels; frequencies are exact powers of two; no pi.         */
/* ------------------------------------------------------------------ */
ORC_API void orc_posenc(const float *x, int64_t M, int C, int L, int include_input, float *out)
{
    const int E = 2 * L * C + (include_input ? C : 0);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        float *e = out + m * E;
        int p = 0;
        if (include_input)
            for (int c = 0; c < C; ++c) e[p++] = x[m * C + c];
        float freq = 1.0f;
        for (int l = 0; l < L; ++l) {
            for (int c = 0; c < C; ++c) e[p++] = sinf(freq * x[m * C + c]);
            for (int c = 0; c < C; ++c) e[p++] = cosf(freq * x[m * C + c]);
            freq = freq * 2.0f;
        }
    }
}

/* ------------------------------------------------------------------ */
/* f4: SHEncoder.encode.  R/signal_encoder/spherical_harmonics_encoder.py:86-139            */
/* Real spherical-harmonics basis up to l = 4 on the UN-normalised input (x, y, z); python     */
/* float coefficients meet fp32 tensors, so every coefficient is rounded to fp32 first and     */
/* every product / sum is a separately rounded fp32 operation in the reference's order.        */
/* ------------------------------------------------------------------ */
static const double SH0 = 0.28209479177387814, SH1 = 0.4886025119029199;
static const double SH2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
                              0.5462742152960396};
static const double SH3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                              -0.4570457994644658, 1.445305721320277, -0.5900435899266435};
static const double SH4[9] = {2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892,
                              0.10578554691520431, -0.6690465435572892, 0.47308734787878004, -1.7701307697799304,
                              0.6258357354491761};

ORC_API void orc_shenc(const float *in, int64_t M, int degree, float *out)
{
    const int E = degree * degree;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const float x = in[3 * m], y = in[3 * m + 1], z = in[3 * m + 2];
        float *e = out + m * E;
        e[0] = (float)SH0;
        if (degree > 1) {
            e[1] = (float)-SH1 * y; e[2] = (float)SH1 * z; e[3] = (float)-SH1 * x;
        }
        if (degree > 2) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            e[4] = (float)SH2[0] * xy;
            e[5] = (float)SH2[1] * yz;
            e[6] = (float)SH2[2] * ((2.0f * zz - xx) - yy);
            e[7] = (float)SH2[3] * xz;
            e[8] = (float)SH2[4] * (xx - yy);
            if (degree > 3) {
                e[9] = ((float)SH3[0] * y) * (3.0f * xx - yy);
                e[10] = ((float)SH3[1] * xy) * z;
                e[11] = ((float)SH3[2] * y) * ((4.0f * zz - xx) - yy);
                e[12] = ((float)SH3[3] * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
                e[13] = ((float)SH3[4] * x) * ((4.0f * zz - xx) - yy);
                e[14] = ((float)SH3[5] * z) * (xx - yy);
                e[15] = ((float)SH3[6] * x) * (xx - 3.0f * yy);
            }
            if (degree > 4) {
                e[16] = ((float)SH4[0] * xy) * (xx - yy);
                e[17] = ((float)SH4[1] * yz) * (3.0f * xx - yy);
                e[18] = ((float)SH4[2] * xy) * (7.0f * zz - 1.0f);
                e[19] = ((float)SH4[3] * yz) * (7.0f * zz - 3.0f);
                e[20] = (float)SH4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f);
                e[21] = ((float)SH4[5] * xz) * (7.0f * zz - 3.0f);
                e[22] = ((float)SH4[6] * (xx - yy)) * (7.0f * zz - 1.0f);
                e[23] = ((float)SH4[7] * xz) * (xx - 3.0f * yy);
                e[24] = (float)SH4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
            }
        }
    }
}

/* basis in double (for the reverse pass below) */
static void sh_basis_d(double x, double y, double z, int degree, double *e)
{
    const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    e[0] = SH0;
    if (degree > 1) { e[1] = -SH1 * y; e[2] = SH1 * z; e[3] = -SH1 * x; }
    if (degree > 2) {
        e[4] = SH2[0] * xy; e[5] = SH2[1] * yz; e[6] = SH2[2] * (2.0 * zz - xx - yy); e[7] = SH2[3] * xz;
        e[8] = SH2[4] * (xx - yy);
    }
    if (degree > 3) {
        e[9] = SH3[0] * y * (3 * xx - yy); e[10] = SH3[1] * xy * z; e[11] = SH3[2] * y * (4 * zz - xx - yy);
        e[12] = SH3[3] * z * (2 * zz - 3 * xx - 3 * yy); e[13] = SH3[4] * x * (4 * zz - xx - yy);
        e[14] = SH3[5] * z * (xx - yy); e[15] = SH3[6] * x * (xx - 3 * yy);
    }
    if (degree > 4) {
        e[16] = SH4[0] * xy * (xx - yy); e[17] = SH4[1] * yz * (3 * xx - yy); e[18] = SH4[2] * xy * (7 * zz - 1);
        e[19] = SH4[3] * yz * (7 * zz - 3); e[20] = SH4[4] * (zz * (35 * zz - 30) + 3);
        e[21] = SH4[5] * xz * (7 * zz - 3); e[22] = SH4[6] * (xx - yy) * (7 * zz - 1);
        e[23] = SH4[7] * xz * (xx - 3 * yy); e[24] = SH4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy));
    }
}

/* reverse of orc_shenc (autograd in the reference): g_in = J^T g_out with the Jacobian of the polynomials taken by
 * central differences of the double-precision basis (they are polynomials of degree <= 4: the difference quotient
 * with h = 1e-3 is exact to ~1e-12 relative; no second hand-derived formula to get wrong) */
ORC_API void orc_shenc_backward(const float *in, const float *g_out, int64_t M, int degree, float *g_in)
{
    const int E = degree * degree;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const double p[3] = {in[3 * m], in[3 * m + 1], in[3 * m + 2]};
        for (int c = 0; c < 3; ++c) {
            /* five-point stencil: exact for polynomials up to degree 4 */
            const double h = 1e-2 * (fabs(p[c]) > 1.0 ? fabs(p[c]) : 1.0);
            double q[3], ep2[25], ep1[25], em1[25], em2[25];
            q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
            q[c] = p[c] + 2 * h; sh_basis_d(q[0], q[1], q[2], degree, ep2);
            q[c] = p[c] + h;     sh_basis_d(q[0], q[1], q[2], degree, ep1);
            q[c] = p[c] - h;     sh_basis_d(q[0], q[1], q[2], degree, em1);
            q[c] = p[c] - 2 * h; sh_basis_d(q[0], q[1], q[2], degree, em2);
            double acc = 0.0;
            for (int k = 0; k < E; ++k)
                acc += (double)g_out[m * E + k] * ((-ep2[k] + 8.0 * ep1[k] - 8.0 * em1[k] + em2[k]) / (12.0 * h));
            g_in[3 * m + c] = (float)acc;
        }
    }
}

/* ------------------------------------------------------------------ */
/* a10: the NeRF MLP.  R/network/nerf.py:24-63 (layers), :102-119.     */
/* Parameters arrive as ONE flat fp32 blob in state_dict order         */
/* fc_in.weight, fc_in.bias, fc_1.weight, ... fc_out.weight,           */
/* fc_out.bias, each weight (out,in) row-major (nn.Linear layout).     */
/* ------------------------------------------------------------------ */
#define NL 11
typedef struct { int in, out; int64_t w_off, b_off; } layer_t;

static void nerf_layout(int E_p, int E_d, int F, layer_t *L, int64_t *total)
{
    const int ins[NL] = {E_p, F, F, F, F, F + E_p, F, F, F, F + E_d, F / 2};
    const int outs[NL] = {F, F, F, F, F, F, F, F, F + 1, F / 2, 3};
    int64_t off = 0;
    for (int l = 0; l < NL; ++l) {
        L[l].in = ins[l]; L[l].out = outs[l];
        L[l].w_off = off; off += (int64_t)ins[l] * outs[l];
        L[l].b_off = off; off += outs[l];
    }
    if (total) *total = off;
}

ORC_API int64_t orc_nerf_param_count(int E_p, int E_d, int F)
{
    layer_t L[NL]; int64_t t; nerf_layout(E_p, E_d, F, L, &t); return t;
}

/* y(out) = b + W x, fp32, k ascending (sequential accumulate) */
static void linear_fwd(const float *W, const float *b, int in, int out, const float *x, float *y)
{
    for (int n = 0; n < out; ++n) {
        const float *w = W + (int64_t)n * in;
        float acc = b[n];
        for (int k = 0; k < in; ++k) acc = acc + x[k] * w[k];
        y[n] = acc;
    }
}

/* per-sample activation record kept for the backward pass */
typedef struct {
    float *x0;            /* E_p       pos enc                                  */
    float *h[8];          /* F each    post-ReLU outputs of fc_in, fc_1..fc_7   */
    float *y8;            /* F+1       fc_8 output (no ReLU)                    */
    float *h9;            /* F/2       post-ReLU fc_9                           */
    float *y10;           /* 3         fc_out pre-sigmoid                       */
} act_t;

static void mlp_forward_one(const float *P, const layer_t *L, int E_p, int E_d, int F,
                            const float *pe, const float *de, act_t *A, float *cat5, float *cat9,
                            float *sigma, float *rgb)
{
    memcpy(A->x0, pe, sizeof(float) * E_p);
    /* fc_in + 4 trunk layers, ReLU each: nerf.py:102-106 */
    linear_fwd(P + L[0].w_off, P + L[0].b_off, E_p, F, pe, A->h[0]);
    for (int k = 0; k < F; ++k) A->h[0][k] = A->h[0][k] > 0.0f ? A->h[0][k] : 0.0f;
    for (int l = 1; l <= 4; ++l) {
        linear_fwd(P + L[l].w_off, P + L[l].b_off, F, F, A->h[l - 1], A->h[l]);
        for (int k = 0; k < F; ++k) A->h[l][k] = A->h[l][k] > 0.0f ? A->h[l][k] : 0.0f;
    }
    /* skip: cat([pos, x]) -- pos FIRST: nerf.py:108 */
    memcpy(cat5, pe, sizeof(float) * E_p);
    memcpy(cat5 + E_p, A->h[4], sizeof(float) * F);
    linear_fwd(P + L[5].w_off, P + L[5].b_off, F + E_p, F, cat5, A->h[5]);
    for (int k = 0; k < F; ++k) A->h[5][k] = A->h[5][k] > 0.0f ? A->h[5][k] : 0.0f;
    for (int l = 6; l <= 7; ++l) {
        linear_fwd(P + L[l].w_off, P + L[l].b_off, F, F, A->h[l - 1], A->h[l]);
        for (int k = 0; k < F; ++k) A->h[l][k] = A->h[l][k] > 0.0f ? A->h[l][k] : 0.0f;
    }
    /* fc_8, NO ReLU; sigma = relu(x[:,0]): nerf.py:113-115 */
    linear_fwd(P + L[8].w_off, P + L[8].b_off, F, F + 1, A->h[7], A->y8);
    *sigma = A->y8[0] > 0.0f ? A->y8[0] : 0.0f;
    /* cat([x[:,1:], view_dir]) -- features FIRST: nerf.py:116 */
    memcpy(cat9, A->y8 + 1, sizeof(float) * F);
    memcpy(cat9 + F, de, sizeof(float) * E_d);
    linear_fwd(P + L[9].w_off, P + L[9].b_off, F + E_d, F / 2, cat9, A->h9);
    for (int k = 0; k < F / 2; ++k) A->h9[k] = A->h9[k] > 0.0f ? A->h9[k] : 0.0f;
    linear_fwd(P + L[10].w_off, P + L[10].b_off, F / 2, 3, A->h9, A->y10);
    for (int c = 0; c < 3; ++c) rgb[c] = 1.0f / (1.0f + expf(-A->y10[c]));
}

static void act_alloc(act_t *A, int E_p, int F)
{
    A->x0 = (float *)malloc(sizeof(float) * E_p);
    for (int l = 0; l < 8; ++l) A->h[l] = (float *)malloc(sizeof(float) * F);
    A->y8 = (float *)malloc(sizeof(float) * (F + 1));
    A->h9 = (float *)malloc(sizeof(float) * (F / 2));
    A->y10 = (float *)malloc(sizeof(float) * 3);
}
static void act_free(act_t *A)
{
    free(A->x0);
    for (int l = 0; l < 8; ++l) free(A->h[l]);
    free(A->y8); free(A->h9); free(A->y10);
}

/* forward on pre-encoded inputs: NeRF.forward(pos (M,E_p), view_dir (M,E_d)) */
ORC_API void orc_mlp_forward(const float *params, int E_p, int E_d, int F, const float *pos_enc,
                             const float *dir_enc, int64_t M, float *sigma, float *rgb)
{
    layer_t L[NL];
    nerf_layout(E_p, E_d, F, L, NULL);
#pragma omp parallel
    {
        act_t A; act_alloc(&A, E_p, F);
        float *cat5 = (float *)malloc(sizeof(float) * (F + E_p));
        float *cat9 = (float *)malloc(sizeof(float) * (F + E_d));
#pragma omp for schedule(static)
        for (int64_t m = 0; m < M; ++m)
            mlp_forward_one(params, L, E_p, E_d, F, pos_enc + m * E_p, dir_enc + m * E_d, &A, cat5,
                            cat9, sigma + m, rgb + 3 * m);
        free(cat5); free(cat9); act_free(&A);
    }
}

/* The values the ten ReLUs of nerf.py:102-118 are applied to, in the mask layout of orc_mlp_backward_ex:
 * pre (M, 8 F + F/2 + 1) = [fc_in .. fc_7 outputs | fc_9 output | fc_8 output[0]].  Tests use it on a handful of
 * samples to find the units whose pre-activation sits within rounding of zero -- the only decisions another fp32
 * summation order (the reference's sgemm, the MFMA chain) can take differently. */
ORC_API void orc_mlp_preacts(const float *params, int E_p, int E_d, int F, const float *pos_enc,
                             const float *dir_enc, int64_t M, float *pre)
{
    layer_t L[NL];
    nerf_layout(E_p, E_d, F, L, NULL);
    const int64_t row_len = 8 * (int64_t)F + F / 2 + 1;
    float *h = (float *)malloc(sizeof(float) * F);
    float *y8 = (float *)malloc(sizeof(float) * (F + 1));
    float *cat5 = (float *)malloc(sizeof(float) * (F + E_p));
    float *cat9 = (float *)malloc(sizeof(float) * (F + E_d));
    for (int64_t m = 0; m < M; ++m) {
        float *row = pre + m * row_len;
        const float *pe = pos_enc + m * E_p, *de = dir_enc + m * E_d;
        for (int l = 0; l <= 7; ++l) {
            if (l == 0) {
                linear_fwd(params + L[0].w_off, params + L[0].b_off, E_p, F, pe, row);
            } else if (l == 5) {
                memcpy(cat5, pe, sizeof(float) * E_p);
                memcpy(cat5 + E_p, h, sizeof(float) * F);
                linear_fwd(params + L[5].w_off, params + L[5].b_off, F + E_p, F, cat5, row + 5 * (int64_t)F);
            } else {
                linear_fwd(params + L[l].w_off, params + L[l].b_off, F, F, h, row + l * (int64_t)F);
            }
            for (int k = 0; k < F; ++k) { const float v = row[l * (int64_t)F + k]; h[k] = v > 0.0f ? v : 0.0f; }
        }
        linear_fwd(params + L[8].w_off, params + L[8].b_off, F, F + 1, h, y8);
        row[8 * (int64_t)F + F / 2] = y8[0];
        memcpy(cat9, y8 + 1, sizeof(float) * F);
        memcpy(cat9 + F, de, sizeof(float) * E_d);
        linear_fwd(params + L[9].w_off, params + L[9].b_off, F + E_d, F / 2, cat9, row + 8 * (int64_t)F);
    }
    free(h); free(y8); free(cat5); free(cat9);
}

/* ------------------------------------------------------------------ */
/* a13 (MLP part): parameter gradients for upstream (g_sigma (M,),     */
/* g_rgb (M,3)).  Hand-derived reverse of nerf.py:102-119.  Sums over   */
/* samples run in double (the reference's autograd sums in fp32 inside */
/* sgemm: a tolerance quantity; the oracle is the yardstick).          */
/* ------------------------------------------------------------------ */
static void linear_bwd(const float *W, int in, int out, const float *x, const float *gy,
                       double *gW, double *gb, float *gx /* may be NULL */, int gx_from, int gx_n)
{
    /* parameter gradients accumulate in double: the oracle is the yardstick */
    for (int n = 0; n < out; ++n) {
        const double g = (double)gy[n];
        if (g == 0.0) continue;
        double *gw = gW + (int64_t)n * in;
        for (int k = 0; k < in; ++k) gw[k] += g * (double)x[k];
        gb[n] += g;
    }
    if (gx) {
        for (int k = 0; k < gx_n; ++k) {
            float acc = 0.0f;
            for (int n = 0; n < out; ++n) acc = acc + gy[n] * W[(int64_t)n * in + gx_from + k];
            gx[k] = acc;
        }
    }
}

/* g_pos (M,E_p) / g_dir (M,E_d): gradients w.r.t. the (encoded) network inputs, what autograd hands back for
 * `pos` / `view_dir` of nerf.py:65-68 (NULL: not wanted).  masks (M, 8 F + F/2 + 1) bytes: 1 where the ReLU of
 * h0..h7 (F each), h9 (F/2) and the density head relu(y8[0]) let the unit through (NULL: not wanted) -- tests diff
 * them against the kernel's mask planes to tell ReLU flips at |pre-activation| < 1 ulp from arithmetic error.
 * force_masks (same layout, NULL: the oracle's own): the backward takes its ReLU derivatives from THESE decisions
 * instead of its own activations -- with the kernel's masks the two backward passes differentiate the same
 * piecewise-linear function, and what remains is summation-order rounding only. */
ORC_API void orc_mlp_backward_ex(const float *params, int E_p, int E_d, int F, const float *pos_enc,
                                 const float *dir_enc, int64_t M, const float *g_sigma,
                                 const float *g_rgb, float *g_params /* zero-filled by caller */,
                                 float *g_pos, float *g_dir, unsigned char *masks,
                                 const unsigned char *force_masks)
{
    layer_t L[NL];
    int64_t total;
    nerf_layout(E_p, E_d, F, L, &total);
    int nthreads = 1;
#if defined(_OPENMP)
    nthreads = omp_get_max_threads();
#endif
    double **local = (double **)calloc((size_t)nthreads, sizeof(double *));
#pragma omp parallel
    {
        int tid = 0;
#if defined(_OPENMP)
        tid = omp_get_thread_num();
#endif
        double *G = (double *)calloc((size_t)total, sizeof(double));
        local[tid] = G;
        act_t A; act_alloc(&A, E_p, F);
        float *cat5 = (float *)malloc(sizeof(float) * (F + E_p));
        float *cat9 = (float *)malloc(sizeof(float) * (F + E_d));
        float *ga = (float *)malloc(sizeof(float) * (F + 1));
        float *gb_ = (float *)malloc(sizeof(float) * (F + 1));
        float *gcat = (float *)malloc(sizeof(float) * (F + (E_p > E_d ? E_p : E_d)));
        float sg, col[3];
#pragma omp for schedule(static)
        for (int64_t m = 0; m < M; ++m) {
            mlp_forward_one(params, L, E_p, E_d, F, pos_enc + m * E_p, dir_enc + m * E_d, &A, cat5,
                            cat9, &sg, col);
            const int64_t mask_row = 8 * (int64_t)F + F / 2 + 1;
            if (masks) {
                unsigned char *mk = masks + m * mask_row;
                for (int l = 0; l < 8; ++l)
                    for (int k = 0; k < F; ++k) mk[l * F + k] = A.h[l][k] > 0.0f;
                for (int k = 0; k < F / 2; ++k) mk[8 * F + k] = A.h9[k] > 0.0f;
                mk[8 * F + F / 2] = A.y8[0] > 0.0f;
            }
            const unsigned char *fm = force_masks ? force_masks + m * mask_row : NULL;
#define ON_H(l, k) (fm ? fm[(l) * F + (k)] != 0 : A.h[l][k] > 0.0f)
#define ON_H9(k) (fm ? fm[8 * F + (k)] != 0 : A.h9[k] > 0.0f)
#define ON_SIGMA (fm ? fm[8 * F + F / 2] != 0 : A.y8[0] > 0.0f)
            /* fc_out: rgb = sigmoid(y10) */
            float gy10[3];
            for (int c = 0; c < 3; ++c) gy10[c] = g_rgb[3 * m + c] * col[c] * (1.0f - col[c]);
            linear_bwd(params + L[10].w_off, F / 2, 3, A.h9, gy10, G + L[10].w_off,
                       G + L[10].b_off, ga, 0, F / 2);
            for (int k = 0; k < F / 2; ++k) ga[k] = ON_H9(k) ? ga[k] : 0.0f;
            /* fc_9 on cat9 = [y8[1:], dir] */
            linear_bwd(params + L[9].w_off, F + E_d, F / 2, cat9, ga, G + L[9].w_off,
                       G + L[9].b_off, gcat, 0, g_dir ? F + E_d : F);
            memcpy(gb_ + 1, gcat, sizeof(float) * F);
            if (g_dir) memcpy(g_dir + m * E_d, gcat + F, sizeof(float) * E_d);
            /* y8[0] -> sigma = relu(y8[0]) */
            gb_[0] = ON_SIGMA ? g_sigma[m] : 0.0f;
            /* fc_8 (no relu) */
            linear_bwd(params + L[8].w_off, F, F + 1, A.h[7], gb_, G + L[8].w_off, G + L[8].b_off,
                       ga, 0, F);
            for (int l = 7; l >= 6; --l) {
                for (int k = 0; k < F; ++k) ga[k] = ON_H(l, k) ? ga[k] : 0.0f;
                linear_bwd(params + L[l].w_off, F, F, A.h[l - 1], ga, G + L[l].w_off,
                           G + L[l].b_off, gb_, 0, F);
                memcpy(ga, gb_, sizeof(float) * F);
            }
            /* fc_5 on cat5 = [pos, h4] */
            for (int k = 0; k < F; ++k) ga[k] = ON_H(5, k) ? ga[k] : 0.0f;
            if (g_pos) {
                linear_bwd(params + L[5].w_off, F + E_p, F, cat5, ga, G + L[5].w_off, G + L[5].b_off,
                           gcat, 0, F + E_p);
                memcpy(g_pos + m * E_p, gcat, sizeof(float) * E_p);
                memcpy(gb_, gcat + E_p, sizeof(float) * F);
            } else {
                linear_bwd(params + L[5].w_off, F + E_p, F, cat5, ga, G + L[5].w_off, G + L[5].b_off,
                           gb_, E_p, F);
            }
            memcpy(ga, gb_, sizeof(float) * F);
            for (int l = 4; l >= 1; --l) {
                for (int k = 0; k < F; ++k) ga[k] = ON_H(l, k) ? ga[k] : 0.0f;
                linear_bwd(params + L[l].w_off, F, F, A.h[l - 1], ga, G + L[l].w_off,
                           G + L[l].b_off, gb_, 0, F);
                memcpy(ga, gb_, sizeof(float) * F);
            }
            for (int k = 0; k < F; ++k) ga[k] = ON_H(0, k) ? ga[k] : 0.0f;
            linear_bwd(params + L[0].w_off, E_p, F, A.x0, ga, G + L[0].w_off, G + L[0].b_off,
                       g_pos ? gcat : NULL, 0, g_pos ? E_p : 0);
            if (g_pos)   /* autograd adds the two contributions to `pos` (fc_5's skip input, fc_in) */
                for (int k = 0; k < E_p; ++k) g_pos[m * E_p + k] = g_pos[m * E_p + k] + gcat[k];
        }
        free(cat5); free(cat9); free(ga); free(gb_); free(gcat); act_free(&A);
    }
    for (int64_t i = 0; i < total; ++i) {
        double s = 0.0;
        for (int t = 0; t < nthreads; ++t)
            if (local[t]) s += local[t][i];
        g_params[i] = (float)((double)g_params[i] + s);
    }
    for (int t = 0; t < nthreads; ++t) free(local[t]);
    free(local);
}

ORC_API void orc_mlp_backward(const float *params, int E_p, int E_d, int F, const float *pos_enc,
                              const float *dir_enc, int64_t M, const float *g_sigma,
                              const float *g_rgb, float *g_params /* zero-filled by caller */)
{
    orc_mlp_backward_ex(params, E_p, E_d, F, pos_enc, dir_enc, M, g_sigma, g_rgb, g_params, NULL, NULL, NULL, NULL);
}

/* ------------------------------------------------------------------ */
/* a11: quadrature integrator.                                         */
/* R/renderer/integrators/quadrature_integrator.py:14-67               */
/* The exclusive prefix sum uses ATen's CPU cumsum semantics (double   */
/* accumulator, each prefix rounded to fp32).                          */
/* ------------------------------------------------------------------ */
ORC_API void orc_composite_forward(const float *sigma, const float *radiance, const float *delta,
                                   int64_t n, int64_t S, float *rgb, float *w)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double run = 0.0;
        float acc[3] = {0.0f, 0.0f, 0.0f};
        for (int64_t s = 0; s < S; ++s) {
            const float tau = sigma[i * S + s] * delta[i * S + s];
            const float T = expf(-(float)run);
            const float alpha = 1.0f - expf(-tau);
            const float wi = T * alpha;
            w[i * S + s] = wi;
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + wi * radiance[(i * S + s) * 3 + c];
            run += (double)tau;
        }
        for (int c = 0; c < 3; ++c) rgb[3 * i + c] = acc[c];
    }
}

/* ------------------------------------------------------------------ */
/* a13 (integrator part): reverse of the quadrature rule.              */
/* With G_i = g_rgb . c_i (+ g_w_i):                                   */
/*   dL/dc_i     = w_i * g_rgb                                         */
/*   dL/dsigma_i = delta_i * (T_{i+1} G_i - sum_{k>i} w_k G_k)         */
/* evaluated in double here (the oracle is the yardstick).             */
/* ------------------------------------------------------------------ */
ORC_API void orc_composite_backward(const float *sigma, const float *radiance, const float *delta,
                                    const float *g_rgb, const float *g_w /* NULL ok */, int64_t n,
                                    int64_t S, float *g_sigma, float *g_radiance)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double *T = (double *)malloc(sizeof(double) * (S + 1));
        double *wv = (double *)malloc(sizeof(double) * S);
        double run = 0.0;
        for (int64_t s = 0; s < S; ++s) {
            const double tau = (double)(sigma[i * S + s] * delta[i * S + s]);
            T[s] = exp(-run);
            wv[s] = T[s] * (1.0 - exp(-tau));
            run += tau;
        }
        T[S] = exp(-run);
        double suffix = 0.0;
        for (int64_t s = S - 1; s >= 0; --s) {
            double G = g_w ? (double)g_w[i * S + s] : 0.0;
            for (int c = 0; c < 3; ++c) {
                G += (double)g_rgb[3 * i + c] * (double)radiance[(i * S + s) * 3 + c];
                g_radiance[(i * S + s) * 3 + c] = (float)(wv[s] * (double)g_rgb[3 * i + c]);
            }
            g_sigma[i * S + s] = (float)((double)delta[i * S + s] * (T[s + 1] * G - suffix));
            suffix += wv[s] * G;
        }
        free(T);
        free(wv);
    }
}

/* ------------------------------------------------------------------ */
/* f1: optimizer step.  The reference builds                           */
/*   torch.optim.Adam(params, lr=init_lr, eps=eps)                     */
/* (runners/runner_utils.py:691-695; default betas, no weight decay,   */
/* no amsgrad) and calls optimizer.step() once per batch               */
/* (runners/train.py:216).  The arithmetic is torch's (third party,    */
/* torch 2.10.0 torch/optim/adam.py _single_tensor_adam -- the CPU     */
/* default), restated in the same operation order; Python-float        */
/* scalars are rounded to fp32 when they meet an fp32 tensor; bias     */
/* corrections and step size are host doubles.  lerp_ is a fused       */
/* multiply-add in ATen's vectorised CPU kernel (aten/src/ATen/native/ */
/* cpu/LerpKernel.cpp lerp_vec), so m is a tolerance quantity.         */
/* `step` counts from 1; lr is the scheduler-decayed rate.             */
/* ------------------------------------------------------------------ */
ORC_API void orc_adam_step(float *p, const float *g, float *m, float *v, int64_t n, int64_t step,
                           double lr, double beta1, double beta2, double eps)
{
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2);
    const float bc2_sqrt = (float)sqrt(bc2), epsf = (float)eps, neg_step = (float)(-(lr / bc1));
    for (int64_t i = 0; i < n; ++i) {
        m[i] = fmaf(w1, g[i] - m[i], m[i]);
        v[i] = v[i] * b2 + (w2 * g[i]) * g[i];
        const float denom = sqrtf(v[i]) / bc2_sqrt + epsf;
        p[i] = p[i] + neg_step * (m[i] / denom);
    }
}

/* ExponentialLR as the reference configures it (runners/runner_utils.py:701-711):          */
/* gamma = pow(end_lr / init_lr, 1 / num_iter); scheduler.step() after every optimizer step */
/* (runners/train.py:217-218).  torch's closed form is not used by step(): it multiplies    */
/* the running lr by gamma each call (torch/optim/lr_scheduler.py ExponentialLR.get_lr).    */
ORC_API double orc_exponential_lr(double init_lr, double end_lr, int64_t num_iter, int64_t steps_done)
{
    const double gamma = pow(end_lr / init_lr, 1.0 / (double)num_iter);
    double lr = init_lr;
    for (int64_t i = 0; i < steps_done; ++i) lr = lr * gamma;
    return lr;
}
