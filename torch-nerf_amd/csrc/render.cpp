// a3 + a12: one render_scene pass over a ray range as a single enqueue
// (sample -> fused encode+MLP -> integrate), replacing the reference's Python loop over
// ray batches (R/renderer/volume_renderer.py:229-254).  Inference only.
#include "common.h"
#include "net.h"

namespace {
inline int64_t align256(int64_t b) { return (b + 255) & ~(int64_t)255; }
}  // namespace

// render_fused.hip: the whole pass as ONE kernel when the sample count tiles (64, 192, ...)
int nerf_render_rays_fused(const mlp::Net &net, const void *packed, const float *ray_o, const float *ray_d, int64_t n, int Sc, int Sf,
                           const float *t_bins, float partition_size, float *weights_in, const float *u1,
                           const float *u2, const float *u3, float *rgb, float *weights_out, int64_t *bin_idx,
                           float *t_out, nerf_stream_t stream);

NERF_API int64_t nerf_render_workspace_bytes(int64_t n, int S) {
    if (n < 0 || S <= 0) return 0;
    // (callers that know the pass is fused -- nerf_render_is_fused -- may pass a null workspace)
    const int64_t m = n * (int64_t)S;
    // pts, dirs (m,3) ; delta, sigma (m) ; radiance (m,3)
    return 3 * align256(m * 12) + 2 * align256(m * 4);
}

NERF_API int nerf_render_pass(const nerf_net_t *net_abi, const void *packed, const float *ray_o, const float *ray_d, int64_t n,
                              int Sc, int Sf, const float *t_bins, float partition_size,
                              float *weights_in, const float *u1, const float *u2, const float *u3,
                              float *rgb, float *weights_out, int64_t *bin_idx, float *t, void *workspace,
                              nerf_stream_t stream) {
    mlp::Net net;
    if (int rc = nerf::fused_net(net_abi, net, "nerf_render_rays")) return rc;
    if (!nerf::raw_inputs_ok(net))
        return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_render_rays: the pass encodes raw sample points: nerf_net_t needs both encode levels");
    NERF_REQUIRE(n >= 0 && Sc > 0 && Sf >= 0, "nerf_render_rays: bad sizes");
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(packed && rgb && weights_out, "nerf_render_rays: null pointer");
    const int S = Sc + (weights_in ? Sf : 0);
    NERF_REQUIRE(ray_o && ray_d && t_bins && u1 && (!weights_in || Sf == 0 || (u2 && u3)),
                 "nerf_render_rays: null pointer");
    if (nerf_render_is_fused(Sc, Sf, weights_in != nullptr))   // one launch, nothing but rgb + weights written
        return nerf_render_rays_fused(net, packed, ray_o, ray_d, n, Sc, Sf, t_bins, partition_size, weights_in, u1, u2,
                                      u3, rgb, weights_out, bin_idx, t, stream);
    NERF_REQUIRE(workspace, "nerf_render_rays: this sample count needs the workspace (see nerf_render_workspace_bytes)");
    const int64_t m = n * (int64_t)S;
    char *ws = static_cast<char *>(workspace);
    float *pts = reinterpret_cast<float *>(ws); ws += align256(m * 12);
    float *dirs = reinterpret_cast<float *>(ws); ws += align256(m * 12);
    float *radiance = reinterpret_cast<float *>(ws); ws += align256(m * 12);
    float *delta = reinterpret_cast<float *>(ws); ws += align256(m * 4);
    float *sigma = reinterpret_cast<float *>(ws);
    int rc;
    if (weights_in)
        rc = nerf_sample_hierarchical(ray_o, ray_d, n, Sc, Sf, t_bins, partition_size, weights_in, u1, u2,
                                      u3, bin_idx, t, pts, dirs, delta, stream);
    else
        rc = nerf_sample_stratified(ray_o, ray_d, n, Sc, t_bins, partition_size, u1, t, pts, dirs,
                                    delta, stream);
    if (rc != NERF_OK) return rc;
    rc = nerf_mlp_forward(net_abi, packed, pts, dirs, m, 0, sigma, radiance, nullptr, stream);
    if (rc != NERF_OK) return rc;
    return nerf_composite_forward(sigma, radiance, delta, n, S, rgb, weights_out, stream);
}

NERF_API int nerf_render_rays(const nerf_net_t *net, const void *packed, const float *ray_o, const float *ray_d, int64_t n,
                              int Sc, int Sf, const float *t_bins, float partition_size,
                              float *weights_in, const float *u1, const float *u2, const float *u3,
                              float *rgb, float *weights_out, void *workspace, nerf_stream_t stream) {
    return nerf_render_pass(net, packed, ray_o, ray_d, n, Sc, Sf, t_bins, partition_size, weights_in, u1, u2, u3, rgb,
                            weights_out, nullptr, nullptr, workspace, stream);
}
