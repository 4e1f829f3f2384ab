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
//  * weights stream L2 -> LDS through a 4-slot ring of 32-KiB chunks filled by LDS-DMA
//    (global_load_lds_dwordx4, issued 3 chunks ahead, counted vmcnt + one s_barrier per
//    chunk), shared by the 4 wavefronts; A fragments come out of LDS with conflict-free
//    ds_read_b128 (XOR-swizzled rows)
//  * positional encodings are computed in registers straight into B-fragment layout
//  * bound: fp32 MFMA (9280 MFMAs = 38.0 MFLOP per 32 samples); HBM traffic 40 B/sample
#include "common.h"
#include "mlp_layout.h"

namespace {

using namespace mlp;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// sin and cos with a 3-term Cody-Waite reduction (FMA) and Cephes minimax polynomials:
// <= ~1.5e-7 abs error for |x| < 3e4; larger arguments take the library path.
__device__ __forceinline__ void sincos_cw(float x, float &s, float &c) {
    if (__builtin_expect(fabsf(x) > 30000.0f, 0)) {
        s = sinf(x);
        c = cosf(x);
        return;
    }
    const float n = rintf(x * 0.636619747f);
    float r = fmaf(-n, 1.57079637e+0f, x);
    r = fmaf(-n, -4.37113883e-8f, r);
    r = fmaf(-n, -1.71512451e-15f, r);
    const int q = (int)n;
    const float r2 = r * r;
    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    cp = fmaf(cp * r2, r2, fmaf(r2, -0.5f, 1.0f));
    const float ss = (q & 1) ? cp : sp;
    const float cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// feature k of PositionalEncoder(3, L, include_input=True).encode((x,y,z)); 0 beyond kmax
// layout (positional_encoder.py:83-88): [x y z | sin(2^0 xyz) cos(2^0 xyz) | sin(2^1 xyz) ...]
__device__ __forceinline__ float enc_feature(int k, float x, float y, float z, int kmax) {
    const int e = k - 3;
    const int f = e / 6;
    const int r6 = e - 6 * f;
    const int ch = k < 3 ? k : (r6 >= 3 ? r6 - 3 : r6);
    const float v = ch == 0 ? x : (ch == 1 ? y : z);
    float s, c;
    sincos_cw(ldexpf(v, f < 0 ? 0 : f), s, c);
    const float t = r6 >= 3 ? c : s;
    return k < 3 ? v : (k < kmax ? t : 0.0f);
}

// one 1-KiB piece per instruction: LDS[m0 + lane*16] <- global[src]
__device__ __forceinline__ void lds_dma_16(const char *src, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_dst)
        : "memory");
}

struct Pipe {
    const char *src_lane;  // stream base + this lane's byte offset inside a chunk
    unsigned lds_wave;     // LDS byte address of ring slot 0 + this wave's offset
    unsigned issued;       // chunks issued so far
    int issue_pos;         // stream position (0..n_chunks-1) of the next chunk to issue
    unsigned consumed;     // chunks consumed so far
    int n_chunks;

    __device__ __forceinline__ void issue() {
        const char *s = src_lane + (size_t)issue_pos * CHUNK_BYTES;
        const unsigned d = lds_wave + (issued & (RING_SLOTS - 1)) * CHUNK_BYTES;
#pragma unroll
        for (int j = 0; j < 8; ++j) lds_dma_16(s + j * 1024, d + j * 1024);
        ++issued;
        issue_pos = (issue_pos + 1 == n_chunks) ? 0 : issue_pos + 1;
    }
    // make the next chunk readable; returns the LDS byte offset (from slot 0) of its image
    __device__ __forceinline__ unsigned acquire() {
        // three chunks (3 x 8 DMA instructions of this wave) are in flight: the oldest must land
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's pieces landed; everyone left the slot refilled next
        asm volatile("" ::: "memory");
        issue();
        const unsigned slot = consumed & (RING_SLOTS - 1);
        ++consumed;
        return slot * CHUNK_BYTES;
    }
};

template <int NFB>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[8], const f32x16 &b, const char *chunk,
                                          const int (&offq)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(chunk + fb * 4096 + offq[q]);
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * q + 0], acc[fb], 0, 0, 0);
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * q + 1], acc[fb], 0, 0, 0);
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * q + 2], acc[fb], 0, 0, 0);
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * q + 3], acc[fb], 0, 0, 0);
        }
    }
}

// acc[fb][4q..4q+3] <- bias[32 fb + 8 q + 4 h ..]: the C fragment starts as the bias
template <int NFB>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[8], const float *bias, int h) {
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(bias + 32 * fb + 8 * q + 4 * h);
            acc[fb][4 * q + 0] = v.x;
            acc[fb][4 * q + 1] = v.y;
            acc[fb][4 * q + 2] = v.z;
            acc[fb][4 * q + 3] = v.w;
        }
}

template <int NFB>
__device__ __forceinline__ void save_plane(float *plane, int width, int64_t m, bool valid, int h,
                                           const f32x16 *blk) {
    if (!valid) return;
    float *row = plane + m * width + 4 * h;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {blk[fb][4 * q], blk[fb][4 * q + 1], blk[fb][4 * q + 2], blk[fb][4 * q + 3]};
            *reinterpret_cast<f32x4 *>(row + 32 * fb + 8 * q) = v;
        }
}

// sum over this lane's half of the features of w[feature] * x[feature]; w in LDS
template <int NFB>
__device__ __forceinline__ float half_dot(const float *w, const f32x16 *x, int h) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(w + 32 * fb + 8 * q + 4 * h);
            s0 = fmaf(v.x, x[fb][4 * q + 0], s0);
            s1 = fmaf(v.y, x[fb][4 * q + 1], s1);
            s2 = fmaf(v.z, x[fb][4 * q + 2], s2);
            s3 = fmaf(v.w, x[fb][4 * q + 3], s3);
        }
    return (s0 + s1) + (s2 + s3);
}

template <bool ENCODED, bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_forward_kernel(const char *__restrict__ packed,
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
    pipe.src_lane = packed + CONST_BYTES + wave * 8192 + lane * 16;
    pipe.lds_wave =
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 8192u;
    pipe.issued = 0;
    pipe.issue_pos = 0;
    pipe.consumed = 0;
    pipe.n_chunks = FWD_CHUNKS;
    __syncthreads();
    pipe.issue();
    pipe.issue();
    pipe.issue();

    const int64_t ntiles = (M + TILE_SAMPLES - 1) / TILE_SAMPLES;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t m = tile * TILE_SAMPLES + wave * 32 + i;
        const bool valid = m < M;
        const int64_t mc = valid ? m : M - 1;

        // ---- encodings, straight into B-fragment layout: reg r <-> feature 32 kb + (r&3) + 8 (r>>2) + 4 h
        f32x16 pe[2], de;
        if (ENCODED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
                pe[0][r] = pos[mc * E_POS + k];
                pe[1][r] = (32 + k < E_POS) ? pos[mc * E_POS + 32 + k] : 0.0f;
                de[r] = (k < E_DIR) ? dir[mc * E_DIR + k] : 0.0f;
            }
        } else {
            const float px = pos[3 * mc], py = pos[3 * mc + 1], pz = pos[3 * mc + 2];
            const float dx = dir[3 * mc], dy = dir[3 * mc + 1], dz = dir[3 * mc + 2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
                pe[0][r] = enc_feature(k, px, py, pz, E_POS);
                pe[1][r] = enc_feature(32 + k, px, py, pz, E_POS);
                de[r] = enc_feature(k, dx, dy, dz, E_DIR);
            }
        }
        float *sv = saved;
        if (SAVE) {
            save_plane<2>(sv, 64, m, valid, h, pe);
            save_plane<1>(sv + M * (64 + 9 * 256 + 128), 32, m, valid, h, &de);
            sv += M * 64;
        }

        f32x16 acc[8], act[8];

        // ---- fc_in + ReLU (nerf.py:102)
        load_bias<8>(acc, cb + CB_BIAS, h);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) mma_chunk<8>(acc, pe[kb], lds + pipe.acquire(), offq);
#pragma unroll
        for (int fb = 0; fb < 8; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) act[fb][r] = fmaxf(acc[fb][r], 0.0f);
        if (SAVE) { save_plane<8>(sv, 256, m, valid, h, act); sv += M * 256; }

        // ---- fc_1 .. fc_8 (nerf.py:103-113); skip connection at fc_5 (pos FIRST, :108)
        float sigma_pre = 0.0f;
        for (int l = 1; l <= 8; ++l) {
            if (l == 8) sigma_pre = half_dot<8>(cb + CB_W8ROW0, act, h);  // density row of fc_8
            load_bias<8>(acc, l < 8 ? cb + CB_BIAS + l * 256 : cb + CB_BIAS8, h);
            if (l == 5) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) mma_chunk<8>(acc, pe[kb], lds + pipe.acquire(), offq);
            }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) mma_chunk<8>(acc, act[kb], lds + pipe.acquire(), offq);
            const float floor_ = l < 8 ? 0.0f : -INFINITY;  // fc_8 has no ReLU (:113)
#pragma unroll
            for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) act[fb][r] = fmaxf(acc[fb][r], floor_);
            if (SAVE) { save_plane<8>(sv, 256, m, valid, h, act); sv += M * 256; }
        }
        sigma_pre += __shfl_xor(sigma_pre, 32, WAVE);
        const float sigma = fmaxf(sigma_pre + cb[CB_SCALARS], 0.0f);  // relu(x[:,0]) (:115)

        // ---- fc_9 on cat([x[:,1:], view_dir]) -- features FIRST (:116-118)
        load_bias<4>(acc, cb + CB_BIAS9, h);
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) mma_chunk<4>(acc, act[kb], lds + pipe.acquire(), offq);
        mma_chunk<4>(acc, de, lds + pipe.acquire(), offq);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fb][r] = fmaxf(acc[fb][r], 0.0f);
        if (SAVE) save_plane<4>(sv, 128, m, valid, h, acc);

        // ---- fc_out + sigmoid (:119) on the vector ALU: 3 x 128 MACs per sample
        float y[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float p = half_dot<4>(cb + CB_WOUT + c * HALF, acc, h);
            p += __shfl_xor(p, 32, WAVE);
            y[c] = 1.0f / (1.0f + expf(-(p + cb[CB_SCALARS + 1 + c])));
        }
        if (valid && h == 0) {
            sigma_out[m] = sigma;
            rgb_out[3 * m + 0] = y[0];
            rgb_out[3 * m + 1] = y[1];
            rgb_out[3 * m + 2] = y[2];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int num_compute_units() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <bool ENCODED, bool SAVE>
int launch_forward(const void *packed, const float *pos, const float *dir, int64_t M, float *sigma,
                   float *rgb, void *saved, hipStream_t stream) {
    auto kern = mlp_forward_kernel<ENCODED, SAVE>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return nerf::check_launch("nerf_mlp_forward: LDS attribute");
        configured = true;
    }
    const int64_t ntiles = (M + TILE_SAMPLES - 1) / TILE_SAMPLES;
    const int cus = num_compute_units();
    const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, stream,
                       reinterpret_cast<const char *>(packed), pos, dir, M, sigma, rgb,
                       reinterpret_cast<float *>(saved));
    return nerf::check_launch("nerf_mlp_forward");
}

}  // namespace

NERF_API int64_t nerf_mlp_saved_bytes(int64_t M) {
    return M < 0 ? 0 : M * (int64_t)mlp::SAVED_FLOATS_PER_SAMPLE * 4;
}

NERF_API int nerf_mlp_forward(const void *packed, const float *pos, const float *view_dir, int64_t M,
                              int encoded, float *sigma, float *rgb, void *saved, nerf_stream_t stream) {
    NERF_REQUIRE(M >= 0, "nerf_mlp_forward: negative M");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(packed && pos && view_dir && sigma && rgb, "nerf_mlp_forward: null pointer");
    hipStream_t s = nerf::as_stream(stream);
    if (encoded)
        return saved ? launch_forward<true, true>(packed, pos, view_dir, M, sigma, rgb, saved, s)
                     : launch_forward<true, false>(packed, pos, view_dir, M, sigma, rgb, saved, s);
    return saved ? launch_forward<false, true>(packed, pos, view_dir, M, sigma, rgb, saved, s)
                 : launch_forward<false, false>(packed, pos, view_dir, M, sigma, rgb, saved, s);
}
