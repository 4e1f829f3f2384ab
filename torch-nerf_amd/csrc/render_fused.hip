// a3 + a6/a7 + a8-a11 + a12 fused: ONE kernel per render_scene pass for inference.
//
// Replaces, for a range of rays, StratifiedSampler.sample_along_rays (R/renderer/ray_samplers/
// stratified_sampler.py:57-128 + utils.py:8-58), PrimitiveCube.query_points (R/scene/primitives/cube.py:39-76 with
// both encoders and NeRF.forward) and QuadratureIntegrator.integrate_along_rays (R/renderer/integrators/
// quadrature_integrator.py:14-67) -- the body of VolumeRenderer._render_ray_batches' loop (volume_renderer.py:
// 229-254) -- and with it every intermediate the reference materialises: sample points and repeated
// directions (stratified_sampler.py:112-126, 24 B/sample), delta, sigma and radiance (20 B/sample) never reach HBM.
// What crosses HBM per ray: o, d, the draws u1 (+ u2, u3), the coarse weights in (floored in place, utils.py:31)
// and weights + pixel colour out.
//
// Structure: the persistent workgroup of mlp_forward.hip (4 wavefronts, LDS-DMA weight ring, activations in
// registers) walks BUNCHES of four rays, one per wavefront, cut into groups of G rays whose G*S samples are a
// whole number of 128-sample tiles (S = 64: 2 rays = 1 tile; S = 192: 2 rays = 3 tiles):
//   1. wavefront w samples ray w of the bunch: the same per-ray code as sampling.hip (render_device.h),
//      sorted positions into an LDS row -- bin indices stay bit-identical to the reference
//   2. per group: its tiles run through mlp::forward_tile; a lane builds its point o + t d in registers; sigma
//      and the colours of the tile go to LDS rows
//   3. per group: the wavefront that sampled a ray integrates it out of those rows (render_device.h: fp64
//      shuffle scan) and writes the ray's weights and pixel colour
// Bound: fp32 MFMA, exactly as mlp_forward.hip; steps 1 and 3 are latency-bound single-wave code and cost ~1 % of a
// bunch's time (cycle budget: scripts/timeline_fused.py).
#include "common.h"
#ifdef X_FUSED_TIMELINE   // scripts/timeline_fused.py: cycle stamps of thread 0 of workgroup 0
__device__ unsigned long long g_stamps[512];
__device__ int g_nstamps;
#define RD_STAMP() do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_nstamps < 512) g_stamps[g_nstamps++] = __builtin_readcyclecounter(); } while (0)
#endif
#include "mlp_device.h"
#include "mlp_tile.h"
#include "render_device.h"
#include "net.h"

namespace {

using namespace mlp;

constexpr int BUNCH = 4;            // rays sampled at once: one per wavefront
constexpr int MAX_GROUP_RAYS = BUNCH;

struct FusedArgs {
    Net net;
    const char *packed;
    const float *ray_o, *ray_d;
    int64_t n;
    int Sc, Sf;              // Sf = 0: coarse pass
    const float *t_bins;
    float ps;
    float *weights_in;       // (n, Sc), floored in place; null for the coarse pass
    const float *u1, *u2, *u3;
    float *rgb, *weights_out;
    int64_t *bin_idx;        // optional (n, Sf)
    float *t_out;            // optional (n, S)
    int G;                   // rays per group
};

constexpr int DIRROW = HALF + 32;   // per ray: fc_9 starting values (128) + the encoded direction (32)

__host__ __device__ inline int fused_lds_floats(int G, int S, int Sc, int Sf) {
    // t: BUNCH rows; sigma G*S, radiance 3*G*S (one group at a time); o,d: 8 per ray (padded); sampling scratch per ray;
    // per ray the fc_9 starting vector + encoded direction
    return BUNCH * S + 4 * G * S + 8 * BUNCH + BUNCH * render::hierarchical_scratch_floats(Sc, Sf) + BUNCH * DIRROW;
}

// fc_9's accumulators after the bias and the direction chunk, for ONE ray, exactly as forward_tile's MFMA chain leaves
// them: v[n] = fma(W[n][k1] e[k1], fma(W[n][k0] e[k0], ...bias[n])) with the k-steps in the order the 32x32x2 MFMAs of
// mma_chunk take them -- group q, step j multiplies feature 8q + j (lane half 0) and then 8q + j + 4 (lane half 1) --
// so that the fused pass stays bit-identical to the kernel chain, whose every sample of the ray computes this same
// vector.  One wavefront per ray: lane k < 32 encodes direction feature k (the same enc_feature / sincos_cw the tile
// uses), every lane then runs the chain for outputs n = lane and lane + 64 with the weights of the stream's direction
// chunk (L2-resident; chunk_slot_offset is its swizzle).
template <int INPUT>
__device__ __forceinline__ void ray_direction_row(const Net &net, const char *packed, const float *cb, const float *od,
                                                  int lane, float *row) {
    float *enc = row + HALF;
    const int e_dir = INPUT == IN_SHIPPED ? DEFAULT_NET.e_dir : net.e_dir;
    const int inc = INPUT == IN_SHIPPED ? 1 : net.inc_dir;
    const int l_dir = INPUT == IN_SHIPPED ? DEFAULT_NET.l_dir : net.l_dir;
    const float d0 = od[4], d1 = od[5], d2 = od[6];
    const float big = fmaxf(fmaxf(fabsf(d0), fabsf(d1)), fabsf(d2));
    const bool exact = !(ldexpf(big, l_dir - 1) < 30000.0f);     // wave-uniform (one ray per wavefront)
    if (lane < 32)
        enc[lane] = exact ? enc_feature<true>(lane, d0, d1, d2, e_dir, inc) : enc_feature<false>(lane, d0, d1, d2, e_dir, inc);
    render::wave_fence();
    const char *chunk = packed + CONST_BYTES + (size_t)CH_FC9_DIR * CHUNK_BYTES;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int n = lane + 64 * half;
        float v = cb[CB_BIAS9 + n];
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int k = 8 * q + j + 4 * hh;
                    const float w = *reinterpret_cast<const float *>(chunk + chunk_slot_offset(n, k >> 2) + 4 * (k & 3));
                    v = fmaf(w, enc[k], v);
                }
        row[n] = v;
    }
}

template <int INPUT>
__global__ __launch_bounds__(256, 1) void render_fused_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float *cb = reinterpret_cast<float *>(lds + RING_SLOTS * CHUNK_BYTES);
    const bool fine = a.weights_in != nullptr;
    const int S = a.Sc + (fine ? a.Sf : 0);
    const int G = a.G, GS = G * S;
    float *t_rows = cb + CONST_FLOATS;          // [BUNCH][S] sorted sample positions
    float *sig_rows = t_rows + BUNCH * S;       // [G*S]      one group at a time
    float *rad_rows = sig_rows + GS;            // [G*S][3]
    float *od_rows = rad_rows + 3 * GS;         // [BUNCH][8]: o, d
    float *scratch = od_rows + 8 * BUNCH;       // [BUNCH][hierarchical_scratch_floats]
    const int scratch_stride = render::hierarchical_scratch_floats(a.Sc, a.Sf);
    float *dir_rows = scratch + BUNCH * scratch_stride;   // [BUNCH][DIRROW]: fc_9 starting vector + encoded direction

    for (int e = tid; e < CONST_FLOATS / 4; e += 256)
        reinterpret_cast<f32x4 *>(cb)[e] = reinterpret_cast<const f32x4 *>(a.packed)[e];

    int offq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) offq[q] = chunk_slot_offset(i, 2 * q + h);

    Pipe pipe;
    pipe.src_wave = a.packed + CONST_BYTES + wave * 8192;
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 8192u;
    pipe.issued = 0;
    pipe.issue_pos = 0;
    pipe.consumed = 0;
    pipe.n_pairs = FWD_CHUNKS / 2;
    pipe.skip_mask = 1ull << FC9_DIR_PAIR;     // the direction's contribution comes per ray (ray_direction_row)
    __syncthreads();
    pipe.issue();

    const int64_t bunches = (a.n + BUNCH - 1) / BUNCH;
    const int tiles_per_group = GS / TILE_SAMPLES, groups_per_bunch = BUNCH / G;
    Timeline tl;

    for (int64_t bunch = blockIdx.x; bunch < bunches; bunch += gridDim.x) {
        RD_STAMP();
        // ---- 1. sampling: wavefront w handles ray w of the bunch
        const int64_t ray = bunch * BUNCH + wave;
        const bool has_ray = ray < a.n;
        {
            float *t_row = t_rows + wave * S;
            if (has_ray) {
                if (lane < 3) {
                    od_rows[8 * wave + lane] = a.ray_o[3 * ray + lane];
                    od_rows[8 * wave + 4 + lane] = a.ray_d[3 * ray + lane];
                }
                if (fine)
                    render::hierarchical_ray(lane, a.Sc, a.Sf, a.t_bins, a.ps, a.weights_in + ray * a.Sc,
                                             a.u1 + ray * a.Sc, a.u2 + ray * a.Sf, a.u3 + ray * a.Sf,
                                             a.bin_idx ? a.bin_idx + ray * a.Sf : nullptr,
                                             scratch + wave * scratch_stride, t_row);
                else
                    render::stratified_ray(lane, S, a.t_bins, a.ps, a.u1 + ray * S, t_row);
                if (a.t_out)
                    for (int s = lane; s < S; s += WAVE) a.t_out[ray * S + s] = t_row[s];
            } else {   // a ragged last bunch: finite filler, never written out
                if (lane < 8) od_rows[8 * wave + lane] = 0.0f;
                for (int s = lane; s < S; s += WAVE) t_row[s] = 0.0f;
            }
            render::wave_fence();   // this wavefront's own o, d row is complete
            ray_direction_row<INPUT>(a.net, a.packed, cb, od_rows + 8 * wave, lane, dir_rows + wave * DIRROW);
        }
        RD_STAMP();
        __syncthreads();   // rows of this bunch are complete; the previous bunch's last integration has finished
        RD_STAMP();

        for (int grp = 0; grp < groups_per_bunch; ++grp) {
            if (bunch * BUNCH + grp * G >= a.n) break;   // (wave-uniform) nothing but filler left
            // ---- 2. the group's samples through the fused encode + MLP, 128 at a time
            for (int tile = 0; tile < tiles_per_group; ++tile) {
                const int ml = tile * TILE_SAMPLES + wave * 32 + i;   // sample index inside the group
                const int r = grp * G + ml / S;                       // its ray inside the bunch
                float raw[6];
                {
                    const float t = t_rows[grp * GS + ml];
                    const float *od = od_rows + 8 * r;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        raw[c] = __fadd_rn(od[c], __fmul_rn(t, od[4 + c]));   // stratified_sampler.py:112-117
                        raw[3 + c] = od[4 + c];                                 // :120-126 (directions repeated)
                    }
                }
                float sigma, y[3];
                forward_tile<INPUT, false, true>(a.net, raw, nullptr, nullptr, 0, 0, 0, h, pipe, lds, cb, offq, nullptr,
                                                 []() {}, tl, sigma, y, dir_rows + r * DIRROW);
                if (h == 0) {
                    sig_rows[ml] = sigma;
                    rad_rows[3 * ml + 0] = y[0];
                    rad_rows[3 * ml + 1] = y[1];
                    rad_rows[3 * ml + 2] = y[2];
                }
            }
            RD_STAMP();
            __syncthreads();   // sigma / radiance rows of the group complete
            RD_STAMP();

            // ---- 3. integration by the wavefront that sampled the ray
            const int wl = wave - grp * G;   // this wave's ray inside the group, if it has one
#ifdef X_FUSED_NOCOMP
            if (has_ray && wl >= 0 && wl < G && a.n < 0) {
#else
            if (has_ray && wl >= 0 && wl < G) {
#endif
                const float *t_row = t_rows + wave * S;
                const float *sg = sig_rows + wl * S;
                const float *cl = rad_rows + 3 * wl * S;
                float out[3];
                render::composite_ray(
                    lane, S, [&](int s) { return sg[s]; },
                    [&](int s) { return __fsub_rn(s + 1 < S ? t_row[s + 1] : 1e8f, t_row[s]); },   // stratified_sampler.py:99-105
                    [&](int s, int c) { return cl[3 * s + c]; }, a.weights_out + ray * S, out);
                if (lane == 0) {
                    a.rgb[3 * ray + 0] = out[0];
                    a.rgb[3 * ray + 1] = out[1];
                    a.rgb[3 * ray + 2] = out[2];
                }
            }
            RD_STAMP();
            if (grp + 1 < groups_per_bunch) __syncthreads();   // the rows are free for the next group's tiles
        }
    }
    RD_STAMP();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// rays per group such that G * S is a whole number of tiles; 0 if the fused pass does not apply
int group_rays(int S, int Sc, int Sf) {
    for (int G = 1; G <= MAX_GROUP_RAYS; G *= 2)   // G divides BUNCH
        if ((G * S) % TILE_SAMPLES == 0)
            return (LDS_BYTES + 4 * fused_lds_floats(G, S, Sc, Sf) <= 160 * 1024) ? G : 0;
    return 0;
}

}  // namespace

// 1 if nerf_render_rays runs (n, Sc, Sf) as ONE kernel, 0 if it falls back to the three-kernel chain
NERF_API int nerf_render_is_fused(int Sc, int Sf, int fine) {
    if (Sc <= 0 || Sf < 0) return 0;
    return group_rays(Sc + (fine ? Sf : 0), Sc, fine ? Sf : 0) > 0;
}

int nerf_render_rays_fused(const mlp::Net &net, const void *packed, const float *ray_o, const float *ray_d, int64_t n, int Sc, int Sf,
                           const float *t_bins, float partition_size, float *weights_in, const float *u1,
                           const float *u2, const float *u3, float *rgb, float *weights_out, int64_t *bin_idx,
                           float *t_out, nerf_stream_t stream) {
    const int S = Sc + (weights_in ? Sf : 0);
    FusedArgs a;
    a.net = net;
    a.packed = static_cast<const char *>(packed);
    a.ray_o = ray_o; a.ray_d = ray_d; a.n = n; a.Sc = Sc; a.Sf = weights_in ? Sf : 0;
    a.t_bins = t_bins; a.ps = partition_size; a.weights_in = weights_in;
    a.u1 = u1; a.u2 = u2; a.u3 = u3; a.rgb = rgb; a.weights_out = weights_out; a.bin_idx = bin_idx; a.t_out = t_out;
    a.G = group_rays(S, Sc, a.Sf);
    if (a.G == 0) return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_render_rays_fused: sample count does not tile");
    const int lds_bytes = LDS_BYTES + 4 * fused_lds_floats(a.G, S, Sc, a.Sf);
    static nerf::DeviceMask configured[2] = {{0}, {0}};
    const bool shipped = net.is_default();   // the reference's shipped yaml: compile-time encoding table
    auto kern = shipped ? render_fused_kernel<IN_SHIPPED> : render_fused_kernel<IN_LEVELS>;
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024, configured[shipped],
                                          "nerf_render_rays: LDS attribute"))
        return rc;
    const int64_t bunches = (n + BUNCH - 1) / BUNCH;
    const int cus = nerf::device_cus();
    hipLaunchKernelGGL(kern, dim3((unsigned)(bunches < cus ? bunches : cus)), dim3(256), lds_bytes,
                       nerf::as_stream(stream), a);
    return nerf::check_launch("nerf_render_rays (fused)");
}

#ifdef X_FUSED_TIMELINE
NERF_API int nerf_debug_stamps(unsigned long long *out, int reset) {
    int n = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_nstamps), sizeof(int));
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 512);
    if (reset) { int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nstamps), &z, sizeof(int)); }
    return n;
}
#endif
