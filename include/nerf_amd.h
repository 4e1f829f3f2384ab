/*
 * nerf_amd.h -- C ABI of libnerf_amd.so, the MI355X (gfx950) volume-rendering path.
 *
 * The reference (DveloperY0115/torch-NeRF) is pure Python/PyTorch: it has NO FFI,
 * plugin or operator interface of its own.  The boundary a
 * maintainer binds is therefore the Python class surface of torch_nerf/src/...,
 * and this header is the native layer directly beneath it: one entry point per
 * reference function on the hot path, each citing the function it replaces
 * (R/ = /root/reference/torch_nerf/src/).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *    buffers are owned by the caller and must outlive the call; no ownership moves
 *  - all tensors are dense, row-major, fp32 unless typed otherwise
 *  - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised
 *  - return value: 0 = NERF_OK, otherwise an error code; nothing throws;
 *    nerf_amd_last_error() returns a thread-local message for the last failure
 *  - re-entrant; no global mutable state
 *  - IEEE fp32 with separately rounded multiply/add wherever the reference's
 *    result decides an integer (sample bins) -- see DESIGN.md "numerics"
 */
#ifndef NERF_AMD_H
#define NERF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_AMD_ABI_VERSION 5

enum {
    NERF_OK = 0,
    NERF_ERR_ARG = 1,         /* null pointer / size out of range                */
    NERF_ERR_UNSUPPORTED = 2, /* configuration outside what the kernels support  */
    NERF_ERR_LAUNCH = 3       /* the HIP runtime rejected a launch               */
};

typedef void *nerf_stream_t; /* hipStream_t */

int nerf_amd_abi_version(void);
const char *nerf_amd_last_error(void);

/* ---- a2: VolumeRenderer._generate_screen_coords, R/renderer/volume_renderer.py:171-190
 * coords[i] = (p % W, (H-1) - p / W) with p = pix ? pix[i] : first + i.  int64 (n,2). */
int nerf_screen_coords(int64_t H, int64_t W, const int64_t *pix, int64_t first, int64_t n,
                       int64_t *coords, nerf_stream_t stream);

/* ---- a4 + a5: RaySamplerBase.generate_rays / _get_ray_directions / map_rays_to_ndc,
 * R/renderer/ray_samplers/sampler_base.py:70-113, :134-197, :199-257
 * Pixel source, first non-null wins: `coords` int64 (n,2) | `pix` int64 (n,) | first+i.
 * intrinsic = (fx, fy, cx, cy) read from the 4x4 matrix; extrinsic_host = 12 floats,
 * the row-major [R|t] 3x4 block of the camera-to-world matrix (HOST memory).
 * project_to_ndc != 0 applies map_rays_to_ndc(focal, z_near, H, W). */
int nerf_generate_rays(const int64_t *coords, const int64_t *pix, int64_t first, int64_t n,
                       int64_t H, int64_t W, float fx, float fy, float cx, float cy,
                       const float *extrinsic_host, int project_to_ndc, double focal,
                       double z_near, float *ray_o, float *ray_d, nerf_stream_t stream);

/* ---- a6: StratifiedSampler.sample_along_rays (coarse branch) + _create_t_bins,
 * R/renderer/ray_samplers/stratified_sampler.py:91-128, :130-164
 * t = t_bins + ps*u1 ; delta = diff(cat[t,1e8]) ; pts = o + t*d ; dirs = d.
 * t_bins (S,) is torch.linspace(t_near,t_far,S+1)[:-1] made by the caller;
 * u1 (n,S) are the caller's torch.rand_like draws.  `t` (n,S) may be NULL. */
int nerf_sample_stratified(const float *ray_o, const float *ray_d, int64_t n, int S,
                           const float *t_bins, float partition_size, const float *u1, float *t,
                           float *pts, float *dirs, float *delta, nerf_stream_t stream);

/* ---- a7: hierarchical branch + sample_pdf,
 * R/renderer/ray_samplers/stratified_sampler.py:57-90, R/renderer/ray_samplers/utils.py:8-58
 * weights (n,Sc) is MUTATED in place (+= 1e-5, utils.py:31).  u1 (n,Sc), u2 (n,Sf),
 * u3 (n,Sf) are the three draws in the reference's order.  Outputs have S = Sc+Sf
 * samples per ray, sorted.  `bin_idx` int64 (n,Sf) and `t` (n,S) may be NULL.
 * Bin indices are bit-exact with the reference's CPU path (ATen sum / cumsum order). */
int nerf_sample_hierarchical(const float *ray_o, const float *ray_d, int64_t n, int Sc, int Sf,
                             const float *t_bins, float partition_size, float *weights,
                             const float *u1, const float *u2, const float *u3, int64_t *bin_idx,
                             float *t, float *pts, float *dirs, float *delta,
                             nerf_stream_t stream);

/* ---- a8: PositionalEncoder.encode, R/signal_encoder/positional_encoder.py:49-104
 * x (M,C) -> out (M, 2*L*C + (include_input ? C : 0)). */
int nerf_posenc(const float *x, int64_t M, int C, int L, int include_input, float *out,
                nerf_stream_t stream);

/* ---- a10: NeRF (11 Linear layers), R/network/nerf.py:24-63, :65-121
 * The network instance every nerf_mlp_* / nerf_render_* call works on: NeRF(pos_dim, view_dir_dim, feat_dim)
 * (nerf.py:24-63) and, for the entries that take RAW points / directions, the two
 * PositionalEncoder(3, levels, include_input) in front of it (R/signal_encoder/positional_encoder.py:27-47) --
 * the values the runners read from yaml (coord_encode_level, dir_encode_level, include_input:
 * R/../runners/runner_utils.py:584-612, R/../configs/signal_encoder/positional_encoding.yaml:2-4).
 * levels < 0: the encoder is not a PositionalEncoder the kernels know; only the pre-encoded entries work.
 * A NULL `net` is the reference's shipped configuration {63, 27, 256, 10, 1, 4, 1}.
 *
 * Two kernel families, chosen by nerf_mlp_path(net):
 *   NERF_PATH_FUSED   feat_dim == 256, pos_dim <= 64, view_dir_dim <= 32: the register-resident persistent
 *                     kernels (nerf_mlp_pack / _forward / _backward / _forward_bf16, nerf_render_*)
 *   NERF_PATH_LAYERED anything else: the same persistent structure with the activations parked in HBM planes
 *                     between layers (nerf_mlp_layered_*), pre-encoded inputs, also returns the input gradients
 * `params` is always the flat state_dict blob: fc_in.weight (feat,pos_dim), fc_in.bias, fc_1.weight, ...
 * fc_out.weight (3,feat/2), fc_out.bias = nerf_mlp_param_count(net) floats. */
typedef struct nerf_net {
    int32_t pos_dim, view_dir_dim, feat_dim;
    int32_t pos_levels, pos_include_input;   /* PositionalEncoder in front of `pos`      (levels < 0: unknown) */
    int32_t dir_levels, dir_include_input;   /* PositionalEncoder in front of `view_dir` (levels < 0: unknown) */
} nerf_net_t;
enum { NERF_PATH_FUSED = 0, NERF_PATH_LAYERED = 1 };
int nerf_mlp_path(const nerf_net_t *net);          /* < 0: invalid description (nerf_amd_last_error) */
int64_t nerf_mlp_param_count(const nerf_net_t *net);

/* NERF_PATH_FUSED: nerf_mlp_pack re-tiles `params` into the LDS image the kernels stream
 * (nerf_mlp_packed_bytes(net) bytes); re-run after every parameter update. */
int64_t nerf_mlp_packed_bytes(const nerf_net_t *net);
int nerf_mlp_pack(const nerf_net_t *net, const float *params, void *packed, nerf_stream_t stream);

/* Forward of PrimitiveCube.query_points (R/scene/primitives/cube.py:39-76) fused with
 * both PositionalEncoder.encode calls and NeRF.forward:
 *   encoded == 0: pos (M,3), view_dir (M,3) raw; encoding happens in registers
 *   encoded != 0: pos (M,63), view_dir (M,27) already encoded (plain NeRF.forward)
 * sigma (M,), rgb (M,3).  `saved` = NULL for inference, or nerf_mlp_saved_bytes(net, M)
 * bytes that receive the activation record nerf_mlp_backward needs. */
int64_t nerf_mlp_saved_bytes(const nerf_net_t *net, int64_t M);
/* Float offset of element (sample m, feature k) inside a plane of `width` (a multiple of 32) features per
 * sample of the activation record / gradient workspace ("TF" layout, csrc/mlp_layout.h): host-side, for
 * tools and tests; the record's planes are otherwise opaque. */
int64_t nerf_mlp_plane_offset(int width, int64_t m, int k);
int nerf_mlp_forward(const nerf_net_t *net, const void *packed, const float *pos, const float *view_dir, int64_t M,
                     int encoded, float *sigma, float *rgb, void *saved, nerf_stream_t stream);

/* ---- a10, bf16 variant (BASELINE configs[2]: "bf16 MLP weights on MFMA"), inference only.
 * Weights and layer inputs are rounded to bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate); bias, ReLU,
 * the density row, fc_out and the sigmoid stay fp32.  pos, view_dir are RAW (M,3).  Parity is a PSNR
 * bound against nerf_mlp_forward, not the 1e-5 bound.  Same `params` blob as nerf_mlp_pack. */
int64_t nerf_mlp_packed_bf16_bytes(const nerf_net_t *net);
int nerf_mlp_pack_bf16(const nerf_net_t *net, const float *params, void *packed_bf16, nerf_stream_t stream);
int nerf_mlp_forward_bf16(const nerf_net_t *net, const void *packed_bf16, const float *pos, const float *view_dir, int64_t M,
                          float *sigma, float *rgb, nerf_stream_t stream);

/* ---- a10, split-f16 variant ("f16x2", ABI v5): NeRF.forward (R/network/nerf.py:102-119) at the fp32 bound -- 1e-5 abs on
 * sigma / rgb, like nerf_mlp_forward -- on the f16 matrix pipe.  Every operand of the ten matrix-pipe layers is split
 * in two f16 parts (weights at pack time, scaled per layer by a power of two; activations in the layer seams) and
 * every k-step forms lo.hi + hi.lo + hi.hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation; encodings, biases,
 * the density row, fc_out and the sigmoid stay fp32.  pos, view_dir are RAW (M,3); inference only; the fused family
 * behind two PositionalEncoders.  Activations beyond +-65504 overflow to inf / NaN (never a silent wrong value).
 * Same `params` blob as nerf_mlp_pack. */
int64_t nerf_mlp_packed_f16x2_bytes(const nerf_net_t *net);
int nerf_mlp_pack_f16x2(const nerf_net_t *net, const float *params, void *packed_f16x2, nerf_stream_t stream);
int nerf_mlp_forward_f16x2(const nerf_net_t *net, const void *packed_f16x2, const float *pos, const float *view_dir, int64_t M,
                           float *sigma, float *rgb, nerf_stream_t stream);
/* The same forward as the TRAINING forward of the fused family: it also writes the activation record of
 * nerf_mlp_forward(..., saved) -- `saved` = nerf_mlp_saved_bytes(net, M) bytes -- which nerf_mlp_backward (fp32 kernels)
 * reads: the forward third of a training step at the split kernel's rate, activations recorded to 2^-22.  Raw points. */
int nerf_mlp_forward_f16x2_record(const nerf_net_t *net, const void *packed_f16x2, const float *pos, const float *view_dir,
                                  int64_t M, float *sigma, float *rgb, void *saved, nerf_stream_t stream);
/* nerf_mlp_backward on the split-f16 kernels: the reverse chain dY(l-1) = W_l^T dY(l) (every sample's gradient carries its
 * own power-of-two scale through the chain: gradients are not O(1)) and the dW GEMMs dW_l = dY_l^T X_l (one power-of-two
 * scale per gradient plane; fp32 accumulation, the fp32 path's partial tiles and fixed-order reduction).  Parameter
 * gradients only; same `saved`, `workspace` and g_params as nerf_mlp_backward. */
int nerf_mlp_backward_f16x2(const nerf_net_t *net, const void *packed, const void *packed_f16x2, int64_t M, const float *sigma,
                            const float *rgb, const void *saved, const float *g_sigma, const float *g_rgb, float *g_params,
                            void *workspace, nerf_stream_t stream);
/* Host-only dry run of the bookkeeping of nerf_mlp_backward (f16x2 = 0) / nerf_mlp_backward_f16x2 (f16x2 = 1) on a device of
 * `cus` compute units (<= 0: 256); nothing launched, no GPU needed: every item's 32-row tiles are covered exactly once by
 * the workgroups the plan gives it, and the partial tiles, bias partials and (f16x2) plane maxima / thin-row partials fit the
 * workspace nerf_mlp_backward_workspace_bytes sized.  NERF_OK, or NERF_ERR_ARG with the failed check in
 * nerf_amd_last_error(). */
int nerf_mlp_backward_plan_check(const nerf_net_t *net, int64_t M, int cus, int f16x2);

/* ---- a13 (MLP part): gradients of all 22 parameter tensors (autograd in the
 * reference, entered at runners/train.py:215).  g_params (param_count floats, same
 * layout as `params`) is OVERWRITTEN.  workspace: nerf_mlp_backward_workspace_bytes(net, M).
 * g_pos (M,pos_dim) / g_view_dir (M,view_dir_dim), each optional (NULL): the gradients autograd returns for the two
 * ENCODED inputs of NeRF.forward (R/network/nerf.py:102, :108, :116) -- whatever `encoded` says; a caller that fed
 * raw points chains nerf_posenc_backward behind them.  Asking for either adds three thin GEMMs to the dX chain. */
int64_t nerf_mlp_backward_workspace_bytes(const nerf_net_t *net, int64_t M);
int nerf_mlp_backward(const nerf_net_t *net, const void *packed, const float *params, const float *pos,
                      const float *view_dir, int64_t M, int encoded, const float *sigma,
                      const float *rgb, const void *saved, const float *g_sigma,
                      const float *g_rgb, float *g_params, float *g_pos, float *g_view_dir,
                      void *workspace, nerf_stream_t stream);

/* ---- a10 + a13, NERF_PATH_LAYERED (any pos_dim / view_dir_dim / feat_dim): NeRF.forward on PRE-ENCODED inputs
 * pos (M,pos_dim), view_dir (M,view_dir_dim) (encoded != 0), or -- like nerf_mlp_forward -- on RAW points / directions
 * (M,3) (encoded == 0; PrimitiveCube.query_points, cube.py:59-76, when nerf_net_t names both PositionalEncoders): the
 * encodings are then written straight into the kernel's input planes, the (M, pos_dim) rows never exist.  ONE persistent launch per forward (and one per reverse chain): the
 * fused family's structure -- 128-sample tiles, weights streamed L2 -> LDS by LDS-DMA from a pre-packed, zero-padded
 * stream, v_mfma_f32_32x32x2_f32 with the weights as the A operand -- with the activations of a tile parked in HBM
 * planes between layers (a network of feat_dim 512 does not fit the register file); bias, ReLU / sigmoid and the two
 * torch.cat of nerf.py:108,:116 are fused; every call packs its streams from `params` first.
 *   keep_record != 0 : the whole batch is recorded (what nerf_mlp_layered_backward needs); requires record_rows >= M
 *   keep_record == 0 : inference, the batch is walked in chunks of <= record_rows rows through the same buffer; networks
 *                      whose activations stay in registers write NO activation planes, need only the two input planes
 *                      of the buffer and take longer chunks (ABI v4: was inferred from record_rows >= M, which made
 *                      every inference call of <= 65536 rows a recording one)
 * record = nerf_mlp_layered_record_bytes(net, record_rows) bytes (constant block + forward stream + planes).
 * Backward = autograd's result for nerf.py:102-119: g_params (OVERWRITTEN, layout of `params`) and, when non-NULL,
 * g_pos (M,pos_dim) / g_view_dir (M,view_dir_dim), the gradients w.r.t. the encoded inputs.  No atomics: the
 * sample-axis reductions are split into fixed slices summed in a fixed order (bit-reproducible gradients).
 * workspace: nerf_mlp_layered_workspace_bytes(net, M).
 * (The record also holds the ReLU decisions of h0..h7 and h9 as bit planes, between fc_8[1:] and h9: what the reverse
 * chain reads.)
 * nerf_mlp_layered_plane: byte offset + padded width of a record plane (0 pos, 1 dir, 2..9 h0..h7, 10 fc_8[1:], 11 h9),
 * for tools and tests. */
int64_t nerf_mlp_layered_record_bytes(const nerf_net_t *net, int64_t rows);
int64_t nerf_mlp_layered_workspace_bytes(const nerf_net_t *net, int64_t M);
int64_t nerf_mlp_layered_plane(const nerf_net_t *net, int64_t rows, int which, int *width);
/* Host-only dry run of nerf_mlp_layered_backward's bookkeeping (ABI v5; nothing launched, no GPU needed): the workspace
 * layout, the dW work list against the budget the workspace was sized for, every window's destination rectangle and
 * read extent, the partial-tile buffer on a device of `cus` compute units (<= 0: 256).  NERF_OK, or NERF_ERR_ARG with
 * the failed check in nerf_amd_last_error(). */
int nerf_mlp_layered_plan_check(const nerf_net_t *net, int64_t M, int cus);
int nerf_mlp_layered_forward(const nerf_net_t *net, const float *params, const float *pos, const float *view_dir,
                             int64_t M, int encoded, float *sigma, float *rgb, void *record, int64_t record_rows,
                             int keep_record, nerf_stream_t stream);
int nerf_mlp_layered_backward(const nerf_net_t *net, const float *params, const float *pos, const float *view_dir,
                              int64_t M, const float *sigma, const float *rgb, const void *record,
                              const float *g_sigma, const float *g_rgb, float *g_params, float *g_pos,
                              float *g_view_dir, void *workspace, nerf_stream_t stream);

/* ---- a8, backward: x (M,C), g_out (M, out_dim) -> g_x (M,C) = the gradient autograd returns for `in_signal` of
 * PositionalEncoder.encode (positional_encoder.py:84-104): g_x = [g_in] + sum_l 2^l (cos(2^l x) g_sin_l - sin(2^l x) g_cos_l). */
int nerf_posenc_backward(const float *x, const float *g_out, int64_t M, int C, int L, int include_input,
                         float *g_x, nerf_stream_t stream);

/* ---- f4 (second encoder family): SHEncoder.encode, R/signal_encoder/spherical_harmonics_encoder.py:86-139 -- what the
 * runners put in front of BOTH network inputs under `signal_encoder: sh` (R/../runners/runner_utils.py:595-604).
 * in_signal (M,3) -> out (M, degree^2), degree 1..5 (the reference defines nothing beyond 25 features), products in
 * the reference's order (bit-identical fp32 values); nerf_shenc_backward: g_out (M, degree^2) -> g_in (M,3), the
 * gradient autograd returns for in_signal.  The network behind it is NeRF(degree^2, degree^2): the pre-encoded
 * entries above with a nerf_net_t whose levels are < 0. */
int nerf_shenc(const float *in_signal, int64_t M, int degree, float *out, nerf_stream_t stream);
int nerf_shenc_backward(const float *in_signal, const float *g_out, int64_t M, int degree, float *g_in,
                        nerf_stream_t stream);

/* ---- a11: QuadratureIntegrator.integrate_along_rays,
 * R/renderer/integrators/quadrature_integrator.py:14-67
 * sigma (n,S), radiance (n,S,3), delta (n,S) -> rgb (n,3), weights (n,S). */
int nerf_composite_forward(const float *sigma, const float *radiance, const float *delta,
                           int64_t n, int S, float *rgb, float *weights, nerf_stream_t stream);

/* ---- a13 (integrator part): reverse of the quadrature rule.
 * g_weights may be NULL (it is in the reference's runners). */
int nerf_composite_backward(const float *sigma, const float *radiance, const float *delta,
                            const float *g_rgb, const float *g_weights, int64_t n, int S,
                            float *g_sigma, float *g_radiance, nerf_stream_t stream);

/* ---- a3 + a12: one VolumeRenderer.render_scene pass (inference) on a ray range,
 * R/renderer/volume_renderer.py:59-169, :192-261 -- the Python batch loop over
 * sample_along_rays (ray_samplers/stratified_sampler.py:57-128) -> query_points (scene/primitives/cube.py:39-76)
 * -> integrate_along_rays (integrators/quadrature_integrator.py:14-67) becomes ONE kernel over `n` rays
 * (csrc/render_fused.hip) whenever the samples of a few rays tile the 128-sample MLP pass
 * (nerf_render_is_fused: S = 64, 128, 192, 256, ...; 32 / 96 / 160 with four rays per group): sample points,
 * directions, delta, sigma and radiance never reach HBM.  Other sample counts run as three launches through
 * `workspace`.
 *   coarse pass: weights_in = NULL, u2 = u3 = NULL, Sf = 0
 *   fine pass:   weights_in (n,Sc) is mutated in place like a7
 * workspace: nerf_render_workspace_bytes(n, Sc+Sf); may be NULL when nerf_render_is_fused(Sc, Sf, fine).
 * nerf_render_pass: the same with the optional outputs of a7 -- bin_idx (n,Sf) int64, t (n,S) sorted sample
 * positions -- for parity checks of the fused kernel (both NULL: identical to nerf_render_rays). */
int nerf_render_is_fused(int Sc, int Sf, int fine);
int64_t nerf_render_workspace_bytes(int64_t n, int S);
int nerf_render_rays(const nerf_net_t *net, const void *packed, const float *ray_o, const float *ray_d, int64_t n, int Sc,
                     int Sf, const float *t_bins, float partition_size, float *weights_in,
                     const float *u1, const float *u2, const float *u3, float *rgb,
                     float *weights_out, void *workspace, nerf_stream_t stream);
int nerf_render_pass(const nerf_net_t *net, const void *packed, const float *ray_o, const float *ray_d, int64_t n, int Sc,
                     int Sf, const float *t_bins, float partition_size, float *weights_in,
                     const float *u1, const float *u2, const float *u3, float *rgb,
                     float *weights_out, int64_t *bin_idx, float *t, void *workspace, nerf_stream_t stream);

/* ---- e: draws for ray-sharded rendering / training.  The reference draws with torch.rand / rand_like inside
 * the sampler (R/renderer/ray_samplers/stratified_sampler.py:77,:109; R/renderer/ray_samplers/utils.py:43,:56);
 * a sharded job needs draws that do not depend on the number of GPUs: out[i] = u(key, first + i), the
 * splitmix64-finaliser stream of torch_nerf.amd.synth.counter_uniform (24-bit mantissa, [0,1)), bit for bit. */
int nerf_counter_uniform(uint64_t key, int64_t first, int64_t count, float *out, nerf_stream_t stream);

/* ---- f1: one torch.optim.Adam step as the reference configures it -- Adam(params, lr, eps) with
 * default betas, no weight decay, no amsgrad (R/../runners/runner_utils.py:691-695), stepped once per
 * batch (R/../runners/train.py:216) -- over ONE flat blob (both networks).  `step` counts from 1;
 * `lr` is the current (scheduler-decayed, runner_utils.py:701-711) rate, a host scalar; grads are
 * multiplied by grad_scale first (1/world after a data-parallel SUM all-reduce, else 1).
 * params, exp_avg, exp_avg_sq are updated in place; the four blobs share one 16-byte phase
 * (address mod 16), e.g. the same element offset into four aligned arenas. */
int nerf_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                   int64_t step, double lr, double beta1, double beta2, double eps, double grad_scale,
                   nerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_AMD_H */
