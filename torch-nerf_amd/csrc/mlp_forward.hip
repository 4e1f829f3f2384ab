// a8 + a9 + a10 fused: positional encoding -> 11-layer NeRF MLP -> (sigma, rgb), forward.
//
// Replaces PrimitiveCube.query_points (R/scene/primitives/cube.py:39-76), both
// PositionalEncoder.encode calls (R/signal_encoder/positional_encoder.py:92-104) and
// NeRF.forward (R/network/nerf.py:102-119) with ONE persistent kernel.
//
// Structure (see mlp_layout.h for the fragment algebra):
//  * workgroup = 4 wavefronts, one per SIMD; each wavefront owns 32 samples and carries
//    their 256 activations in registers (128 VGPRs) through all layers, transposed
//    GEMM  Y^T = W X^T  on v_mfma_f32_32x32x2_f32: the D fragment of one layer IS the
//    B fragment of the next -- activations never touch LDS or HBM
//  * weights stream L2 -> LDS through a ring of two 64-KiB PAIRS of 32-KiB chunks filled by LDS-DMA
//    (global_load_lds_dwordx4): while one pair is consumed (256 MFMAs per wave) the next is in
//    flight, its 16 DMA instructions per wave folded between the MFMAs of the pair's first chunk;
//    one vmcnt wait + one s_barrier per pair, shared by the 4 wavefronts
//  * A fragments come out of LDS with conflict-free ds_read_b128 (XOR-swizzled rows), hand-issued
//    one group ahead into two alternating register buffers
//  * positional encodings are computed in registers straight into B-fragment layout (branch-free
//    Cody-Waite sincos; a wave-uniform test sends tiles with huge coordinates to the library path)
//  * bound: fp32 MFMA (9280 MFMAs = 38.0 MFLOP per 32 samples); HBM traffic 40 B/sample
#include "common.h"
#include "mlp_device.h"
#include "mlp_tile.h"
#include "net.h"

namespace {

using namespace mlp;

template <int INPUT, bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_forward_kernel(const Net net, const char *__restrict__ packed,
                                                              const float *__restrict__ pos,
                                                              const float *__restrict__ dir, int64_t M,
                                                              float *__restrict__ sigma_out,
                                                              float *__restrict__ rgb_out,
                                                              float *__restrict__ saved) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float *cb = reinterpret_cast<float *>(lds + RING_SLOTS * CHUNK_BYTES);

    // const block -> LDS (once)
    for (int e = tid; e < CONST_FLOATS / 4; e += 256)
        reinterpret_cast<f32x4 *>(cb)[e] = reinterpret_cast<const f32x4 *>(packed)[e];

    int offq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) offq[q] = chunk_slot_offset(i, 2 * q + h);

    Pipe pipe;
    pipe.src_wave = packed + CONST_BYTES + wave * 8192;
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave =
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 8192u;
    pipe.issued = 0;
    pipe.issue_pos = 0;
    pipe.consumed = 0;
    pipe.n_pairs = FWD_CHUNKS / 2;
    pipe.skip_mask = 0;
    __syncthreads();
    pipe.issue();

    const int64_t ntiles = (M + TILE_SAMPLES - 1) / TILE_SAMPLES;
    const int64_t MP = padded_rows(M);

    // raw inputs of the first tile; later tiles are prefetched one tile ahead
    float raw[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_raw = [&](int64_t tile) {
        if (INPUT == IN_ENCODED || tile >= ntiles) return;
        int64_t mm = tile * TILE_SAMPLES + wave * 32 + i;
        if (mm >= M) mm = M - 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            raw[c] = pos[3 * mm + c];
            raw[3 + c] = dir[3 * mm + c];
        }
    };
    load_raw(blockIdx.x);

    Timeline tl;
#ifdef X_TIMELINE   // scripts/timeline.py: cycle stamps of workgroup 0, written behind the 3 M colours
    tl.buf = reinterpret_cast<unsigned long long *>(rgb_out + 3 * M);
    tl.n = 0;
    tl.on = blockIdx.x == 0 && tid == 0;
#endif
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        tl.stamp();
        const int64_t m = tile * TILE_SAMPLES + wave * 32 + i;
        const bool valid = m < M;
        const int64_t mc = valid ? m : M - 1;

        float sigma, y[3];
        forward_tile<INPUT, SAVE>(net, raw, pos, dir, mc, m, MP, h, pipe, lds, cb, offq, saved,
                                    [&]() { load_raw(tile + gridDim.x); },  // next tile's points: a whole tile of MFMAs hides the latency
                                    tl, sigma, y);
        if (valid && h == 0) {
            sigma_out[m] = sigma;
            rgb_out[3 * m + 0] = y[0];
            rgb_out[3 * m + 1] = y[1];
            rgb_out[3 * m + 2] = y[2];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int INPUT, bool SAVE>
int launch_forward(const Net &net, const void *packed, const float *pos, const float *dir, int64_t M, float *sigma,
                   float *rgb, void *saved, hipStream_t stream) {
    auto kern = mlp_forward_kernel<INPUT, SAVE>;
    static nerf::DeviceMask configured{0};  // one per template instance
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), LDS_BYTES, configured,
                                          "nerf_mlp_forward: LDS attribute"))
        return rc;
    const int64_t ntiles = (M + TILE_SAMPLES - 1) / TILE_SAMPLES;
    const int cus = nerf::device_cus();
    const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, stream, net,
                       reinterpret_cast<const char *>(packed), pos, dir, M, sigma, rgb,
                       reinterpret_cast<float *>(saved));
    return nerf::check_launch("nerf_mlp_forward");
}

}  // namespace

NERF_API int64_t nerf_mlp_plane_offset(int width, int64_t m, int k) {
    if (width <= 0 || width % 32 || m < 0 || k < 0 || k >= width) return -1;
    return mlp::tf_offset(width, m, k);
}

NERF_API int64_t nerf_mlp_saved_bytes(const nerf_net_t *net, int64_t M) {
    mlp::Net n;
    if (nerf::fused_net(net, n, "nerf_mlp_saved_bytes") != NERF_OK) return -1;
    return M < 0 ? 0 : mlp::padded_rows(M) * (int64_t)mlp::SAVED_BYTES_PER_SAMPLE;
}

NERF_API int nerf_mlp_forward(const nerf_net_t *net_abi, const void *packed, const float *pos, const float *view_dir,
                              int64_t M, int encoded, float *sigma, float *rgb, void *saved, nerf_stream_t stream) {
    mlp::Net net;
    if (int rc = nerf::fused_net(net_abi, net, "nerf_mlp_forward")) return rc;
    NERF_REQUIRE(M >= 0, "nerf_mlp_forward: negative M");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(packed && pos && view_dir && sigma && rgb, "nerf_mlp_forward: null pointer");
    hipStream_t s = nerf::as_stream(stream);
    if (encoded)
        return saved ? launch_forward<IN_ENCODED, true>(net, packed, pos, view_dir, M, sigma, rgb, saved, s)
                     : launch_forward<IN_ENCODED, false>(net, packed, pos, view_dir, M, sigma, rgb, saved, s);
    if (!nerf::raw_inputs_ok(net))
        return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_mlp_forward: raw inputs need both encode levels in nerf_net_t");
    if (net.is_default())   // the reference's shipped yaml: compile-time encoding table
        return saved ? launch_forward<IN_SHIPPED, true>(net, packed, pos, view_dir, M, sigma, rgb, saved, s)
                     : launch_forward<IN_SHIPPED, false>(net, packed, pos, view_dir, M, sigma, rgb, saved, s);
    return saved ? launch_forward<IN_LEVELS, true>(net, packed, pos, view_dir, M, sigma, rgb, saved, s)
                 : launch_forward<IN_LEVELS, false>(net, packed, pos, view_dir, M, sigma, rgb, saved, s);
}
