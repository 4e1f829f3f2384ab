// a13 (MLP part): gradients of the 22 parameter tensors of NeRF(pos_dim <= 64, view_dir_dim <= 32, 256).
//
// The reference obtains them from autograd (entered at runners/train.py:215, per layer
// mm x2 + threshold_backward + sum).  Here the reverse pass is three hand-written stages:
//
//  1. dX chain   (mlp_bwd_dx_kernel): same register-resident structure as the forward --
//     32 samples per wavefront, dY^T(l-1) = W_l^T dY^T(l) on v_mfma_f32_32x32x2_f32 with the
//     TRANSPOSED weight chunks streamed through the LDS ring -- ReLU masks come from the
//     bit planes the training-mode forward saved; every layer's pre-activation gradient is
//     written once to a tile-fragment plane (mlp_layout.h).  8704 MFMAs per 32 samples.  It also leaves the
//     [sample][4] plane of colour gradients and per-wavefront sums of the four scalar output gradients.
//  2. dW GEMMs   (mlp_bwd_dw_kernel): dW_l = dY_l^T X_l with the sample axis as the MFMA
//     reduction dimension.  A workgroup owns a work interval of the concatenated (layer, tile) items, keeps
//     the whole 256 x K_l tile in accumulators (256 AGPRs), and streams 32-row tiles of dY and X --
//     contiguous in HBM -- into a double-buffered LDS stage by LDS-DMA.  The layer bias gradients ride along
//     (the A fragments ARE the dY values to be summed), and so do the two thin outer products an MFMA block
//     would waste -- the density row of fc_8 on the fc_8 item, fc_out (3 x 128) on the fc_9 item -- as a
//     few vector FMAs per k-step.  2 KB of HBM reads per sample-layer for 131 kFLOP: 64 FLOP/B, MFMA-bound.
//  3. a deterministic reduction (mlp_bwd_reduce_kernel) of all per-workgroup partial tiles and the dX chain's
//     scalar sums into the flat gradient blob (state_dict layout).  No atomics anywhere: gradients are
//     reproducible bit for bit.
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "mlp_device.h"
#include "net.h"

namespace {

using namespace mlp;

// ------------------------------------------------------------------------------------------
// stage 1: dX chain
// ------------------------------------------------------------------------------------------
// v where the ReLU of feature (fb, r) was active, else +0: the mask bit, sign-extended to 0 / ~0 by ONE
// v_bfe_i32, ANDed onto the value.  (test + compare + select is three instructions per value plus the wait
// states gfx950 wants between a VCC write and its vector reader, all of it exposed in the layer seam.)
__device__ __forceinline__ float masked(const u32x4 &mk, int fb, int r, float v) {
    int keep;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(mk[fb >> 1]), "n"(16 * (fb & 1) + r));
    return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & keep);
}

// IG: also the gradients w.r.t. the two (encoded) inputs of the network -- what autograd returns for `pos` and
// `view_dir` of NeRF.forward (nerf.py:102, :108, :116) -- from three extra pairs of the transposed stream:
//   g_view_dir = dY9 W9[:, 256:]           behind the fc_9 stage (accumulator block: act[4..7] are idle there)
//   g_pos      = dY0 W_in + dY5 W5[:, :E_p]   in the epilogue (dY5 re-read from its plane; autograd adds the two too)
// +576 MFMAs per 32 samples (+6.6 %).  Both leave as TF planes like every other gradient plane (the chain has no
// registers left for row-major addresses); input_grad_rows_kernel writes the callers' row-major (M, E_p) / (M, E_d).
template <bool IG>
__global__ __launch_bounds__(256, 1) void mlp_bwd_dx_kernel(const char *__restrict__ packed, int64_t M,
                                                             const float *__restrict__ sigma,
                                                             const float *__restrict__ rgb,
                                                             const float *__restrict__ g_sigma,
                                                             const float *__restrict__ g_rgb,
                                                             const float *__restrict__ saved,
                                                             float *__restrict__ dy,
                                                             float *__restrict__ bias_partial) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float *cb = reinterpret_cast<float *>(lds + RING_SLOTS * CHUNK_BYTES);
    for (int e = tid; e < CONST_FLOATS / 4; e += 256)
        reinterpret_cast<f32x4 *>(cb)[e] = reinterpret_cast<const f32x4 *>(packed)[e];

    int offq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) offq[q] = chunk_slot_offset(i, 2 * q + h);
    // sums over this lane's samples of the four scalar output gradients: d y10[0..2] (-> fc_out.bias) and
    // d sigma' (-> fc_8.bias[0]); one partial per wavefront, summed in a fixed order by the reduction kernel
    float bsum4[4] = {0.f, 0.f, 0.f, 0.f};

    Pipe pipe;
    pipe.src_wave = packed + BWD_OFFSET + wave * 8192;
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave =
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 8192u;
    pipe.issued = 0;
    pipe.issue_pos = IG ? 0 : BW_FC9T / 2;   // the first pair of the walk
    pipe.consumed = 0;
    pipe.n_pairs = BWD_CHUNKS / 2;
    pipe.skip_mask = IG ? 0ull : BWD_INPUT_GRAD_PAIRS;
    __syncthreads();
    pipe.issue();

    const int64_t MP = padded_rows(M);
    const u32x4 *masks = reinterpret_cast<const u32x4 *>(saved + pl_masks(MP));
    const int64_t ntiles = MP / TILE_SAMPLES;
#ifdef X_TIMELINE   // scripts/timeline_dx.py: cycle stamps of workgroup 0, written behind the dY planes
    // the tail of the dW partial-tile area (sized for 525 slices, about half are used)
    unsigned long long *tl = reinterpret_cast<unsigned long long *>(
        dy + (MP * DY_FLOATS_PER_SAMPLE + 255) / 256 * 256 + (int64_t)(2 * 256 + 13) * (256 * 256 + 256) - 65536);
    int tl_n = 0;
#define TS() do { if (blockIdx.x == 0 && tid == 0) tl[tl_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define TS() do {} while (0)
#endif
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        TS();
        const int64_t m = tile * TILE_SAMPLES + wave * 32 + i;
        const bool valid = m < M;
        const int64_t mc = valid ? m : M - 1;

        // (IG: hipcc hoists the ~50 per-lane LDS addresses of the constant-block reads below out of the tile loop
        // and, with the chain's 500 registers taken, spills them; an offset it cannot see through keeps the address
        // arithmetic -- a handful of v_add per tile -- inside the loop.  The plain chain is left as it was tuned.)
        const float *cbt = cb;
        if (IG) {
            int zero;
            asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
            cbt = cb + zero;
        }
        // fc_out + sigmoid (nerf.py:119): d y10 = g_rgb * rgb * (1 - rgb)
        // (all eight loads first, from the clamped row: written as `valid ? g_rgb[..] * .. : 0` they became five
        // load-then-wait sequences under exec masks, 12 k cycles of HBM latency per tile)
        float gy[3], yv[3], gv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { yv[c] = rgb[3 * mc + c]; gv[c] = g_rgb[3 * mc + c]; }
        const float sg = sigma[mc], gsg = g_sigma[mc];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = gv[c] * yv[c] * (1.0f - yv[c]);
            gy[c] = valid ? t : 0.0f;
        }
        // sigma = relu(y8[0]) (nerf.py:115)
        const float dsig = (valid && sg > 0.0f) ? gsg : 0.0f;
        if (h == 0) {   // the fc_out weight gradient is summed beside the fc_9 GEMM of the dW kernel, from this plane
            const f32x4 g4 = {gy[0], gy[1], gy[2], 0.0f};
            *reinterpret_cast<f32x4 *>(dy + gy_plane(MP) + 4 * m) = g4;
            bsum4[0] += gy[0]; bsum4[1] += gy[1]; bsum4[2] += gy[2]; bsum4[3] += dsig;
        }

        f32x16 acc[8], act[8];

        // Every stage below acquires its first weight pair BEFORE the vector-ALU section that
        // produces its inputs (mask, gradient-plane stores): the stores then have a whole pair of
        // MFMAs to drain before the next acquire's vmcnt(0) would wait for them.

        // ---- dY9 = (W_out^T d y10) . [h9 > 0]   (vector ALU, 3 x 128 MACs per sample)
        // ReLU masks are fetched one stage ahead of their use: a 16-byte global load issued inside the seam that
        // needs it costs its whole latency there (no MFMA is in flight to cover it)
        u32x4 mk = masks[(int64_t)8 * MP * 2 + 2 * m + h];
        TS();
        const char *w = lds + pipe.acquire();
        TS();
        {
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k0 = 32 * fb + 8 * q + 4 * h;
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(cbt + CB_WOUT + k0);
                    const f32x4 w1 = *reinterpret_cast<const f32x4 *>(cbt + CB_WOUT + HALF + k0);
                    const f32x4 w2 = *reinterpret_cast<const f32x4 *>(cbt + CB_WOUT + 2 * HALF + k0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = fmaf(w2[j], gy[2], fmaf(w1[j], gy[1], w0[j] * gy[0]));
                        act[fb][4 * q + j] = masked(mk, fb, 4 * q + j, v);
                    }
                }
        }
        // (plane stores ride between the MFMA groups of the pairs that multiply the stored blocks: PlaneStore, mlp_device.h)
        PlaneStore st9;
        st9.open(dy + dy9_plane(MP), 128, m, h, act);
        mk = masks[(int64_t)7 * MP * 2 + 2 * m + h];   // for the seam of l = 7
        TS();

        if (IG) {   // g_view_dir = W9[:, 256:]^T dY9 (nerf.py:116: the direction is the tail of fc_9's input)
            mma_slots<1, 4, 0, 16, true>(acc, act, w, offq, &pipe);
            pipe.issue_done();
            save_plane<1, true>(dy + gd_plane(MP), 32, m, h, acc);
            w = lds + pipe.acquire();
        }
        // ---- d y8[1:257] = W9[:, 0:256]^T dY9   (fc_9 input is cat([x[:,1:], dir]): nerf.py:116)
        mma_pair<8, true, 8>(acc, act[0], act[1], w, offq, pipe, &st9, 0);   // accumulators start from C = 0
        w = lds + pipe.acquire();
        mma_pair<8, false, 8>(acc, act[2], act[3], w, offq, pipe, &st9, 8);

        // ---- l = 8 .. 1:  dY(l-1) = (W_l^T dY(l)) . [h(l-1) > 0]
        TS();
        for (int l = 8; l >= 1; --l) {
            TS();
            w = lds + pipe.acquire();
            TS();
            // finish the previous stage: its accumulators are dY(l) before masking
            if (l == 8) {
#pragma unroll
                for (int fb = 0; fb < 8; ++fb) act[fb] = acc[fb];  // fc_8 has no ReLU
                if (h == 0) dy[dsig_plane(MP) + m] = dsig;
            } else {
#pragma unroll
                for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) act[fb][r] = masked(mk, fb, r, acc[fb][r]);
                mk = masks[(int64_t)(l - 1) * MP * 2 + 2 * m + h];   // h(l-1): next seam (l = 1: the dY0 epilogue)
            }
            PlaneStore st;
            st.open(dy + dy_plane(MP, l), 256, m, h, act);
            if (l == 8) {  // the density row of fc_8 contributes w8[0, k] * d y8[0]
#pragma unroll
                for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 wv = *reinterpret_cast<const f32x4 *>(cbt + CB_W8ROW0 + 32 * fb + 8 * q + 4 * h);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[fb][4 * q + j] = wv[j] * dsig;
                    }
            }
            TS();
            if (l == 8) mma_pair<8, false, 8>(acc, act[0], act[1], w, offq, pipe, &st, 0);
            else mma_pair<8, true, 8>(acc, act[0], act[1], w, offq, pipe, &st, 0);   // accumulators start from C = 0
            TS();
#pragma unroll
            for (int pr = 1; pr < 4; ++pr) {
                w = lds + pipe.acquire();
                mma_pair<8, false, 8>(acc, act[2 * pr], act[2 * pr + 1], w, offq, pipe, &st, 8 * pr);
            }
        }
        TS();
        // ---- dY0: mask with h0 and store (no further propagation: the encodings carry no gradient)
        {
#pragma unroll
            for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) act[fb][r] = masked(mk, fb, r, acc[fb][r]);
            save_plane<8, true>(dy + dy_plane(MP, 0), 256, m, h, act);
        }
        if (IG) {   // g_pos = W_in^T dY0 (nerf.py:102) + W5[:, :E_p]^T dY5 (the skip connection, :108: pos first)
            w = lds + pipe.acquire();
            mma_slots<2, 4, 0, 16, true>(acc, act, w, offq, &pipe);
            mma_slots<2, 4>(acc, act + 4, w + CHUNK_BYTES, offq);
            pipe.issue_done();
            // dY5 back from its plane: every lane re-reads the 32 groups it stored four layers ago (save_plane's slots)
            const float *p5 = dy + dy_plane(MP, 5) + (m - i) * 256;
#pragma unroll
            for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(p5 + (fb * 4 + q) * 256 + 4 * ((2 * i + h) ^ (2 * q)));
                    act[fb][4 * q + 0] = v.x; act[fb][4 * q + 1] = v.y; act[fb][4 * q + 2] = v.z; act[fb][4 * q + 3] = v.w;
                }
            w = lds + pipe.acquire();
            mma_slots<2, 4, 0, 16>(acc, act, w, offq, &pipe);
            mma_slots<2, 4>(acc, act + 4, w + CHUNK_BYTES, offq);
            pipe.issue_done();
            save_plane<2, true>(dy + gp_plane(MP), 64, m, h, acc);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = bsum4[c];   // lanes of half 1 hold zeros
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
        if (lane == 0) bias_partial[((int64_t)blockIdx.x * 4 + wave) * 4 + c] = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// TF planes of the input-gradient dX chain -> the callers' row-major tensors (HBM-bound, 0.4 KB/sample):
//   g_pos[m][k] = GP[m][k] (k < E_p),  g_dir[m][k] = GD[m][k] (k < E_d)
__global__ void input_grad_rows_kernel(const float *__restrict__ dy, int64_t M, int e_pos, int e_dir,
                                       float *__restrict__ g_pos, float *__restrict__ g_dir) {
    const int64_t MP = padded_rows(M);
    const int64_t total = M * (int64_t)(e_pos + e_dir);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < M * e_pos) {
            const int64_t m = e / e_pos;
            const int k = (int)(e - m * e_pos);
            if (g_pos) g_pos[e] = dy[gp_plane(MP) + tf_offset(64, m, k)];
        } else {
            const int64_t e2 = e - M * e_pos, m = e2 / e_dir;
            const int k = (int)(e2 - m * e_dir);
            if (g_dir) g_dir[e2] = dy[gd_plane(MP) + tf_offset(32, m, k)];
        }
    }
}

// ------------------------------------------------------------------------------------------
// stage 2: dW GEMMs
// ------------------------------------------------------------------------------------------
constexpr int MAX_GEMMS = 13;
constexpr int BIAS_PARTIAL_FLOATS = 1024 * 4 * 4;   // dX chain: up to 1024 workgroups x 4 waves x 4 sums
constexpr int SLICE_EXTRA = 768;  // floats reserved after each partial tile: bias partial [256], density row [256], sum of dsig [1]
constexpr int DW_LDS_BYTES = 131072 + 1024;  // 2 stages of 64 KiB (wide X) or 3 stages of 40 KiB (thin X), + side-job rows
enum { FLAG_BIAS = 1, FLAG_DENSITY = 2, FLAG_FCOUT = 4 };   // side jobs of the fc_8 / fc_9 items (dw_body)

struct GemmDesc {
    const char *a_src;    // dY plane window: tile t at a_src + t * a_stride, 32 x a_width floats contiguous
    const char *x_src;    // X plane window, likewise
    int64_t a_stride, x_stride;   // bytes between 32-sample tiles (= 128 x the plane's width; a window narrower than
                                  // its plane is a run of whole 4-KiB feature blocks inside the tile)
    int64_t partial_off;  // float offset of slice 0 in the partial buffer
    int64_t w_off;        // destination: weight tensor offset in the flat gradient
    int64_t b_off;        // destination: bias offset (row0 already applied by the reducer)
    int a_width;          // 256 | 128  (output features of the layer = rows of dW)
    int x_width;          // 256 | 64 | 32 (padded input features of this column block)
    int first_block, num_slices;  // the workgroups [first_block, first_block + num_slices) hold tiles of this item
    int cost;                     // relative time of one 32-row tile of this item (work units)
    int64_t unit_off;             // work units of all earlier items: tile j of this item starts at unit_off + j * cost
    int flags;
    int in_features;      // row stride of the destination weight tensor
    int col0, valid_cols; // destination column block
    int row0;             // destination row offset (1 for fc_8: row 0 is the density row)
    int valid_rows;       // rows of the window that exist in the destination (a_width for the fused family)
    int a_split;          // 1 | 2 | 4: the dY window is a_width / a_split wide and the tile's k-steps are split between
                          // a_split groups of wavefronts, whose partial tiles the reducer folds (narrow layers)
};
struct GemmTable {
    GemmDesc g[MAX_GEMMS];
    int n;
    int64_t work_total;           // sum over items of tiles * cost
    int64_t off_b8, off_wout, off_bout;   // flat-gradient offsets of fc_8.bias, fc_out.weight, fc_out.bias
};
// the same for the layered family, whose item lists do not fit a kernel argument: header + items in device memory
struct GemmList {
    int n;
    int64_t work_total;
    // round 6: the side jobs of the fc_8 / fc_9 items (density row and fc_out weight gradients summed beside the GEMM, from
    // tiles that are in LDS anyway) for 256-feature layered networks: the planes they read and where fc_out.weight sits
    const float *side_h9, *side_dsig, *side_gy;
    int64_t off_wout;
};

__device__ __forceinline__ int64_t slice_stride(const GemmDesc &g) {
    return (int64_t)g.a_width * g.x_width + SLICE_EXTRA;
}

__device__ __forceinline__ float lds_read_b32(unsigned lds_addr, int imm_offset) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(imm_offset));
    return v;
}
__device__ __forceinline__ f32x4 lds_read_b128(unsigned lds_addr, int imm_offset) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(imm_offset));
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(N) : "memory");
}

// DENSITY (the fc_8 item, X = h7): the density row of fc_8 rides along on the vector ALU --
//   dW8[0][k] = sum_m dsig[m] h7[m][k],  db8[0] = sum_m dsig[m]   (nerf.py:113-115: sigma = relu(fc_8(x)[:, 0]))
// -- with the X fragments this GEMM holds in registers anyway.  Vector instructions between the MFMAs of a wave are
// NOT free on this machine (measured: ~7 cycles each next to 64-cycle fp32 MFMAs), so the job is kept to two FMAs
// and one LDS read per k-step: wave w takes feature blocks 2w, 2w+1 (the body is compiled once per wave: a run-time
// choice of fragments becomes branches in the k-loop), and dsig reaches the lanes through a 128-byte LDS row that
// wave 0 fills one tile ahead from a HAND-ISSUED load (a compiler-visible load makes hipcc insert vmcnt waits
// that also wait for the tile DMA: 16.8 instead of 8.8 ms; scalar loads at the point of use stall every k-step).
// The sum of dsig itself (db8[0]) comes from the dX chain's per-wavefront sums.  (Round 1 summed
// the row in a separate HBM-bound kernel that read the whole h7 plane, 1 KB/sample, a second time: 0.2 ms per step.)
// FWAVE (the fc_9 item, A = dY9, X = y8): the fc_out weight gradient rides along the same way --
//   dWout[c][k] = sum_m gy[m][c] h9[m][k]   (3 x 128; nerf.py:119: rgb = sigmoid(fc_out(h9)))
// -- h9 is the one activation plane no GEMM reads, so its 32-sample tiles (16 KiB) join the item's DMA stream as a third
// operand; the three colour gradients come from the [sample][4] GY plane the dX chain wrote, through a 512-byte LDS
// row filled one tile ahead by wave 0.  Wave w takes features 32 w .. 32 w + 31: one ds_read_b32 + one ds_read_b128 +
// three v_fmac per k-step.  (Rounds 1-2a summed it in a separate HBM-bound kernel: 0.2 ms per step.)
// ASPLIT > 1 (layers narrower than 128 outputs, NA = 1): the dY window is 128 / ASPLIT wide; wave w takes row block
// w % (4 / ASPLIT) and the k-steps (sample pairs) of group w / (4 / ASPLIT) -- every wavefront does useful MFMAs and the
// dY tile is fetched at its own width (a 64-wide layer as a 128-wide window cost twice the MFMAs and 1.5 x the bytes).
// The groups' partial tiles land in the row blocks of the usual [128][XW] partial; reduce_item folds them.  These
// items are HBM-bound (F / 4 FLOP per byte): more, smaller stages keep ~64 KiB of tiles in flight per CU.
template <int NA, int KB, int DWAVE = -1, int FWAVE = -1, int ASPLIT = 1>   // DWAVE / FWAVE >= 0: side job, compiled for that wave
__device__ __forceinline__ void dw_body(const GemmDesc &g, int slice, int64_t t0, int64_t t1,
                                        const float *__restrict__ saved, const float *__restrict__ dy,
                                        float *__restrict__ partial, int64_t MP, char *lds, int tid, int lane,
                                        int wave) {
    constexpr int AW = 128 * NA;  // a_width: each of the 4 waves owns 32*NA rows
    constexpr int XW = 32 * KB;
    constexpr int AE = AW / ASPLIT;                    // width of the dY window in memory
    constexpr int RB = 4 / ASPLIT, NS = 16 / ASPLIT;   // row blocks (ASPLIT > 1) and k-steps per wave and tile
    static_assert(ASPLIT == 1 || (NA == 1 && DWAVE < 0 && FWAVE < 0), "split-k serves the plain narrow items");
    constexpr int A_PIECES = 32 * AE * 4 / 1024 / 4;  // 1-KiB DMA pieces per wave
    constexpr int X_PIECES = 32 * XW * 4 / 1024 / 4;  // per wave (X tile = 4 | 8 | 32 pieces)
    constexpr bool FCOUT = FWAVE >= 0;
    constexpr int H9_PIECES = FCOUT ? 32 * HALF * 4 / 1024 / 4 : 0;   // per wave: the 16-KiB h9 tile
    constexpr int A_BYTES = 32 * AE * 4, X_BYTES = 32 * XW * 4, H9_BYTES = FCOUT ? 32 * HALF * 4 : 0;
    constexpr int STAGE_BYTES = A_BYTES + X_BYTES + H9_BYTES;
    // thin X tiles finish their MFMAs faster than one DMA round trip: keep two tiles in flight
    constexpr int NSTAGE = ASPLIT > 1 ? (KB <= 2 ? 6 : KB == 4 ? 4 : 3) : (KB <= 2) ? 3 : 2;
    static_assert(NSTAGE * STAGE_BYTES <= DW_LDS_BYTES, "stage ring exceeds the LDS allocation");
    static_assert(A_PIECES + X_PIECES + H9_PIECES <= 16, "one DMA piece per k-step");
    constexpr int PER_WAVE = A_PIECES + X_PIECES + H9_PIECES;  // DMA instructions per wave per tile
    constexpr int PPS = (PER_WAVE + NS - 1) / NS;              // ... issued per k-step
    static_assert((NSTAGE - 2) * PER_WAVE < 64, "counted wait");
    const int rb = ASPLIT > 1 ? wave % RB : wave, kgrp = ASPLIT > 1 ? wave / RB : 0;   // wave-uniform
    const unsigned k_off = (unsigned)(64 * NS * kgrp);         // byte offset of this group's first k-step in a fragment row
    const int i = lane & 31, h = lane >> 5;
    const int frag_base = (i >> 3) * 256 + 4 * ((2 * h + ((i >> 2) & 1)) ^ (2 * ((i >> 3) & 1))) + (i & 3);
    const int frag_swing = 16 * (i >> 4);
    const bool want_bias = (g.flags & FLAG_BIAS) != 0;   // tiles [t0, t1) of the item; an empty range writes a zero partial
    const char *a_src = g.a_src;   // wave-uniform; the lane offset rides in the DMA instruction's vector operand
    const char *x_src = g.x_src;
    const int64_t a_stride = g.a_stride, x_stride = g.x_stride;
    const char *h9_src = reinterpret_cast<const char *>(saved + pl_h9(MP));
    const unsigned lane_off = (unsigned)lane * 16u;
    float *out = partial + g.partial_off + slice * slice_stride(g);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;

    f32x16 acc[NA][KB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bsum[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) bsum[a] = 0.0f;
    constexpr bool DENSITY = DWAVE >= 0;
    static_assert(!DENSITY || (KB == 8 && NSTAGE == 2), "the density side job belongs to the 256-wide fc_8 item");
    static_assert(NSTAGE * STAGE_BYTES + 256 <= DW_LDS_BYTES || !DENSITY, "no room for the dsig rows");
    float drow[2] = {0.0f, 0.0f};
    const float *dsig = dy + dsig_plane(MP);
    // two 128-byte LDS rows behind the stages: row (t & 1) holds dsig of tile t
    const unsigned ds_rows = lds_base + NSTAGE * STAGE_BYTES;
    float ds_pend = 0.0f;   // wave 0: dsig[tile * 32 + i] of the tile AFTER the next one being staged
    static_assert(!FCOUT || (NA == 1 && KB == 8 && NSTAGE == 2 && !DENSITY), "the fc_out side job belongs to the fc_9 item");
    float wout[3] = {0.0f, 0.0f, 0.0f};
    const float *gyp = dy + gy_plane(MP);
    f32x4 gy_pend = {0.f, 0.f, 0.f, 0.f};   // wave 0: gy[tile * 32 + i] of the tile AFTER the next one being staged
    // two 512-byte LDS rows behind the stages: row (t & 1) holds gy[.][0..3] of tile t
    if (FCOUT && FWAVE == 0 && t0 < t1) {
        f32x4 first;
        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(first) : "v"(gyp + (t0 * 32 + i) * 4) : "memory");
        if (h == 0) asm volatile("ds_write_b128 %0, %1" : : "v"(ds_rows + (unsigned)(t0 & 1) * 512u + 16u * i), "v"(first) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gy_pend) : "v"(gyp + ((t0 + 1 < t1 ? t0 + 1 : t1 - 1) * 32 + i) * 4) : "memory");
    }
    if (DENSITY && DWAVE == 0 && t0 < t1) {
        // tile t0 goes straight into its row (visible after the first tile-top barrier), tile t0 + 1 is requested
        float first;
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(first) : "v"(dsig + t0 * 32 + i) : "memory");
        if (h == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(ds_rows + (unsigned)(t0 & 1) * 128u + 4u * i), "v"(first) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(ds_pend) : "v"(dsig + (t0 + 1 < t1 ? t0 + 1 : t1 - 1) * 32 + i) : "memory");
    }

    // piece j (0 .. PER_WAVE-1) of tile t into stage `buf`: A pieces first, then X pieces
    auto issue_piece = [&](int64_t t, int buf, int j) {
        const unsigned ad = lds_base + buf * STAGE_BYTES;
        if (j < A_PIECES) {
            lds_dma_16s(a_src + t * a_stride + (wave + 4 * j) * 1024, lane_off, ad + (wave + 4 * j) * 1024);
        } else if (j < A_PIECES + X_PIECES) {
            const int jx = j - A_PIECES;
            lds_dma_16s(x_src + t * x_stride + (wave + 4 * jx) * 1024, lane_off, ad + A_BYTES + (wave + 4 * jx) * 1024);
        } else {
            const int jh = j - A_PIECES - X_PIECES;
            lds_dma_16s(h9_src + t * H9_BYTES + (wave + 4 * jh) * 1024, lane_off, ad + A_BYTES + X_BYTES + (wave + 4 * jh) * 1024);
        }
    };
    auto issue = [&](int64_t t, int buf) {
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) issue_piece(t, buf, j);
    };

    // prologue: NSTAGE-1 tiles in flight (tiles past the end are re-reads of the last tile: the
    // instruction count per step stays fixed so the counted wait below is exact)
    if (t0 < t1)
        for (int d = 0; d < NSTAGE - 1; ++d) issue(t0 + d < t1 ? t0 + d : t1 - 1, d);
    int buf = 0;
    for (int64_t t = t0; t < t1; ++t) {
        wait_vmcnt<(NSTAGE - 2) * PER_WAVE>();   // tile t landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();            // ... and everybody else's; stage (t-1) is free again
        asm volatile("" ::: "memory");
        // the tile NSTAGE-1 ahead goes into the stage everybody just left: its DMA pieces are issued one per
        // k-step below, between the MFMAs, instead of as a 1 k-cycle burst in front of them
        const int64_t tn_raw = t + NSTAGE - 1;
        const int64_t tn = tn_raw < t1 ? tn_raw : t1 - 1;
        int nbuf = buf + NSTAGE - 1;
        if (nbuf >= NSTAGE) nbuf -= NSTAGE;
        // Fragments out of the TF-layout tiles (mlp_layout.h): lane (i, h) of k-step s wants sample 2s + h,
        // feature 32 FB + i, which sits at float FB*1024 + (i>>3)*256 + 16*(s ^ (i>>4)) + 4*((2h + ((i>>2)&1)) ^ (2*((i>>3)&1)))
        // + (i&3) of its tile: one per-lane base for even and one for odd s, plus compile-time offsets.
        // The reads are issued by hand one k-step ahead of the MFMAs that consume them (16 MFMAs = 1 k cycles
        // cover the LDS latency) with the feature block and the k-step in the instruction's 16-bit offset:
        // left to hipcc they become ds_read2_b32 with an address add each, all at the top of the loop body.
        const unsigned stage = lds_base + (unsigned)(buf * STAGE_BYTES);
        const unsigned a_addr[2] = {stage + k_off + 4u * (unsigned)(rb * NA * 1024 + frag_base + frag_swing),
                                    stage + k_off + 4u * (unsigned)(rb * NA * 1024 + frag_base - frag_swing)};
        const unsigned x_addr[2] = {stage + k_off + (unsigned)A_BYTES + 4u * (unsigned)(frag_base + frag_swing),
                                    stage + k_off + (unsigned)A_BYTES + 4u * (unsigned)(frag_base - frag_swing)};
        if (DENSITY && DWAVE == 0) {
            // behind the tile-top vmcnt(0) the values of tile t + 1 have landed: into the row nobody reads during
            // tile t (readers of row (t+1)&1 = row (t-1)&1 passed this tile's barrier); then request tile t + 2
            if (h == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(ds_rows + (unsigned)((t + 1) & 1) * 128u + 4u * i), "v"(ds_pend) : "memory");
            const int64_t t2 = t + 2 < t1 ? t + 2 : t1 - 1;
            asm volatile("global_load_dword %0, %1, off" : "=v"(ds_pend) : "v"(dsig + t2 * 32 + i) : "memory");
        }
        if (FCOUT && FWAVE == 0) {   // same hand-off for the GY rows of the fc_out side job
            if (h == 0) asm volatile("ds_write_b128 %0, %1" : : "v"(ds_rows + (unsigned)((t + 1) & 1) * 512u + 16u * i), "v"(gy_pend) : "memory");
            const int64_t t2 = t + 2 < t1 ? t + 2 : t1 - 1;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gy_pend) : "v"(gyp + (t2 * 32 + i) * 4) : "memory");
        }
        const unsigned ds_addr = ds_rows + (unsigned)(t & 1) * 128u + 4u * (unsigned)h;
        const unsigned gy_addr = ds_rows + (unsigned)(t & 1) * 512u + 16u * (unsigned)h;
        const unsigned h9_addr[2] = {stage + (unsigned)(A_BYTES + X_BYTES) + 4u * (unsigned)(frag_base + frag_swing),
                                     stage + (unsigned)(A_BYTES + X_BYTES) + 4u * (unsigned)(frag_base - frag_swing)};
        float a[2][NA], b[2][KB], dval[2] = {0.0f, 0.0f}, hval[2] = {0.0f, 0.0f};
        f32x4 gyv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#define DW_FETCH(S)                                                                                   \
        {                                                                                             \
            _Pragma("unroll") for (int nb = 0; nb < NA; ++nb)                                         \
                a[(S) & 1][nb] = lds_read_b32(a_addr[(S) & 1], nb * 4096 + 64 * (S));                 \
            _Pragma("unroll") for (int kb = 0; kb < KB; ++kb)                                         \
                b[(S) & 1][kb] = lds_read_b32(x_addr[(S) & 1], kb * 4096 + 64 * (S));                 \
            if (DENSITY) dval[(S) & 1] = lds_read_b32(ds_addr, 8 * (S));   /* dsig of sample 2 S + h */  \
            if (FCOUT) {                                                                              \
                hval[(S) & 1] = lds_read_b32(h9_addr[(S) & 1], (FCOUT ? FWAVE : 0) * 4096 + 64 * (S)); \
                gyv[(S) & 1] = lds_read_b128(gy_addr, 32 * (S));            /* gy of sample 2 S + h */ \
            }                                                                                         \
        }
        DW_FETCH(0)
        // dW += dY^T X over the 32 samples of the tile: 16 k-steps of 2 samples.  The A fragments
        // (dY values) double as the bias-gradient summands: db[n] = sum over samples of dY[m][n].
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            lds_fragments_ready();   // k-step s has landed
            if (s + 1 < NS) DW_FETCH(s + 1)
#pragma unroll
            for (int p = 0; p < PPS; ++p)
                if (s * PPS + p < PER_WAVE) issue_piece(tn, nbuf, s * PPS + p);
#pragma unroll
            for (int nb = 0; nb < NA; ++nb) {
                bsum[nb] += a[s & 1][nb];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][nb], b[s & 1][kb], acc[nb][kb], 0, 0, 0);
            }
            if (DENSITY) {   // this lane's k-step sample is 2 s + h
                // asm volatile: a plain fmaf is free to sink below the NEXT step's hand-issued fragment reads, whose
                // destination registers the compiler believes valid from the moment of issue (hipcc did exactly
                // that -- all 16 FMAs at the end of the tile, reading fragments in flight; scripts/audit_asm_loads.py
                // caught it)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(drow[j]) : "v"(dval[s & 1]), "v"(b[s & 1][2 * (DENSITY ? DWAVE : 0) + j]));
            }
            if (FCOUT) {   // (asm volatile for the same reason)
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(wout[0]) : "v"(gyv[s & 1].x), "v"(hval[s & 1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(wout[1]) : "v"(gyv[s & 1].y), "v"(hval[s & 1]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(wout[2]) : "v"(gyv[s & 1].z), "v"(hval[s & 1]));
            }
        }
#undef DW_FETCH
        buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
    }
    // the side jobs' last hand-issued prefetch is never consumed: its destination registers are dead from here on and
    // hipcc re-uses them at once -- the load has had a whole tile to land, but nothing SAYS so (scripts/audit_asm_loads.py)
    if (DENSITY || FCOUT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // partial tile of this slice: row-major [AW][XW], then bias[AW].  Stores in saddr form -- scalar row base (SALU),
    // ONE lane register (4 h XW + i) * 4, the feature block in the immediate -- straight out of the accumulator
    // registers.  (hipcc's own version of this loop precomputes 256 64-bit vector addresses, hoists them out of the
    // item loop and spills them: 2.4 KB of scratch per lane.)
    {
        const unsigned lane_bytes = (unsigned)(4 * h * XW + i) * 4u;
#pragma unroll
        for (int nb = 0; nb < NA; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *row = out + (int64_t)(wave * 32 * NA + 32 * nb + (r & 3) + 8 * (r >> 2)) * XW;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    asm volatile("global_store_dword %0, %1, %2 offset:%3"
                                 :
                                 : "v"(lane_bytes), "a"(acc[nb][kb][r]), "s"(row), "n"(kb * 128)
                                 : "memory");
            }
    }
#pragma unroll
    for (int nb = 0; nb < NA; ++nb) {
        const float both = bsum[nb] + __shfl_xor(bsum[nb], 32, WAVE);  // even + odd samples
        if (want_bias && h == 0) out[AW * XW + wave * 32 * NA + 32 * nb + i] = both;
    }
    if (DENSITY) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float both = drow[j] + __shfl_xor(drow[j], 32, WAVE);
            if (h == 0) out[AW * XW + 256 + 32 * (2 * (DENSITY ? DWAVE : 0) + j) + i] = both;
        }
    }
    if (FCOUT) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float both = wout[c] + __shfl_xor(wout[c], 32, WAVE);
            if (h == 0) out[AW * XW + 256 + c * HALF + 32 * (FCOUT ? FWAVE : 0) + i] = both;
        }
    }
}

// LIST: the layered family's list kernel (adds the window shape and the side-job addressing only it needs: the fused
// family's table kernel compiles to the instruction stream it had before round 6)
template <bool LIST>
__device__ __forceinline__ void dw_main(const GemmDesc *items, int n_items, int64_t work_total,
                                        const float *__restrict__ saved_, const float *__restrict__ dy_,
                                        float *__restrict__ partial, int64_t M, char *lds,
                                        const GemmList *side = nullptr) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t MP = padded_rows(M);
    // Workgroup b owns the work interval [W b / B, W (b+1) / B) of the concatenated items (W = sum of tiles x cost):
    // every tile whose start lies in it.  Most workgroups sit inside one item; the ~n workgroups on an item
    // boundary finish one item's tail, flush its partial tile, and go on with the next item's head -- so all B
    // workgroups end within a tile of each other, whatever the items' costs (round 1 gave every item a whole number
    // of equal slices: 251 of 256 CUs busy and 5 % between the first and the last workgroup to finish).
    const int64_t tiles = MP / 32;
    const int64_t B = gridDim.x, b = blockIdx.x;
    const int64_t lo = work_total * b / B, hi = work_total * (b + 1) / B;
    bool first = true;
    for (int k = 0; k < n_items; ++k) {
        const GemmDesc &g = items[k];
        const int slice = (int)b - g.first_block;
        if (slice < 0 || slice >= g.num_slices) continue;
        auto tile_at = [&](int64_t unit) {   // first tile of the item that starts at or after `unit`
            const int64_t rel = unit - g.unit_off;
            const int64_t j = rel <= 0 ? 0 : (rel + g.cost - 1) / g.cost;
            return j < tiles ? j : tiles;
        };
        const int64_t t0 = tile_at(lo), t1 = tile_at(hi);
        if (!first) {   // the previous item's last DMA pieces may still be landing in the stages
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        first = false;
        // side jobs address their planes as saved + pl_h9(MP), dy + dsig_plane(MP), dy + gy_plane(MP) (the fused family's
        // record): a list names the planes themselves, the bases are backed out of them (never dereferenced elsewhere)
        const float *saved = saved_, *dy = dy_;
        if (LIST && side) {
            saved = side->side_h9 - pl_h9(MP);
            dy = (g.flags & FLAG_DENSITY) ? side->side_dsig - dsig_plane(MP) : side->side_gy - gy_plane(MP);
        }
        if (g.flags & FLAG_DENSITY) {
            if (wave == 0) dw_body<2, 8, 0>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
            else if (wave == 1) dw_body<2, 8, 1>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
            else if (wave == 2) dw_body<2, 8, 2>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
            else dw_body<2, 8, 3>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        } else if (g.flags & FLAG_FCOUT) {
            if (wave == 0) dw_body<1, 8, -1, 0>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
            else if (wave == 1) dw_body<1, 8, -1, 1>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
            else if (wave == 2) dw_body<1, 8, -1, 2>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
            else dw_body<1, 8, -1, 3>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        } else if (g.a_width == 256 && g.x_width == 256) dw_body<2, 8>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_width == 256 && g.x_width == 64) dw_body<2, 2>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_width == 128 && g.x_width == 256 && g.a_split <= 1) dw_body<1, 8>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_width == 128 && g.x_width == 32 && g.a_split <= 1) dw_body<1, 1>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
#ifndef NERF_DW_FUSED_SHAPES_ONLY   // the other window shapes of the layered family
        else if (g.a_split == 2 && g.x_width == 256) dw_body<1, 8, -1, -1, 2>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_split == 2 && g.x_width == 128) dw_body<1, 4, -1, -1, 2>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_split == 2 && g.x_width == 64) dw_body<1, 2, -1, -1, 2>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_split == 2) dw_body<1, 1, -1, -1, 2>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_split == 4 && g.x_width == 256) dw_body<1, 8, -1, -1, 4>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_split == 4 && g.x_width == 128) dw_body<1, 4, -1, -1, 4>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_split == 4 && g.x_width == 64) dw_body<1, 2, -1, -1, 4>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_split == 4) dw_body<1, 1, -1, -1, 4>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_width == 256 && g.x_width == 128) dw_body<2, 4>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (LIST && g.a_width == 256 && g.x_width == 96) dw_body<2, 3>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_width == 256 && g.x_width == 32) dw_body<2, 1>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else if (g.a_width == 128 && g.x_width == 128) dw_body<1, 4>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
        else dw_body<1, 2>(g, slice, t0, t1, saved, dy, partial, MP, lds, tid, lane, wave);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256, 1) void mlp_bwd_dw_kernel(GemmTable table, const float *__restrict__ saved,
                                                             const float *__restrict__ dy,
                                                             float *__restrict__ partial, int64_t M,
                                                             unsigned long long *__restrict__ block_clocks) {
    const unsigned long long clk0 = block_clocks ? wall_clock64() : 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    dw_main<false>(table.g, table.n, table.work_total, saved, dy, partial, M, lds);
    if (block_clocks && threadIdx.x == 0) block_clocks[blockIdx.x] = wall_clock64() - clk0;  // 100 MHz ticks
}

// the layered family's item list lives in device memory (header + items)
__global__ __launch_bounds__(256, 1) void mlp_bwd_dw_list_kernel(const GemmList *__restrict__ list,
                                                                  float *__restrict__ partial, int64_t M,
                                                                  unsigned long long *__restrict__ block_clocks) {
    const unsigned long long clk0 = block_clocks ? wall_clock64() : 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const GemmDesc *items = reinterpret_cast<const GemmDesc *>(list + 1);
    dw_main<true>(items, list->n, list->work_total, nullptr, nullptr, partial, M, lds, list->side_h9 ? list : nullptr);
    if (block_clocks && threadIdx.x == 0) block_clocks[blockIdx.x] = wall_clock64() - clk0;  // 100 MHz ticks
}


// ------------------------------------------------------------------------------------------
// stage 2 on the f16 matrix pipe (NeRF.f16x2_training): dW_l = dY_l^T X_l with both operands split in two f16 parts
// ------------------------------------------------------------------------------------------
// Same items, same work partition, same partial-tile format and the same reduction as mlp_bwd_dw_kernel; what changes is
// the inner loop: a 32-sample tile is two k-steps of v_mfma_f32_32x32x16_f16 (16 samples each) instead of sixteen of
// 32x32x2.  An operand fragment is one feature over 8 consecutive samples: eight ds_read_b32 out of the TF-layout tile (the
// XOR of that layout moves a lane's samples around inside a 256-byte run: eight per-lane offsets, computed once), split
// in registers (v_cvt_pk_f16_f32, the residual as one v_fma_mix_f32, v_cvt_pk_f16_f32) and fed to lo.hi + hi.lo + hi.hi.
// The sum runs OVER samples, so dY cannot carry a scale per sample here: it takes ONE power of two per gradient plane, from
// the plane's largest |dY| that the split reverse chain leaves behind (atomicMax, order-independent).  A sample whose
// gradient is 2^15 below the largest is carried to fewer bits -- and weighs 2^-15 in the sum: the error stays below the
// fp32 rounding of the sum (tests/test_gpu_f16x2.py drives eight decades).  X (activations, encodings) is O(1): unscaled.
// The two thin side rows (density row of fc_8, fc_out) are a separate pass here (fused_thin_kernel).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
struct ItemPlanes { signed char p[MAX_GEMMS]; };   // which gradient plane (0..8 = dY0..dY8, 9 = dY9) an item's dY window is

__device__ __forceinline__ void split8(const float (&v)[8], h16x8 &hi, h16x8 &lo) {
    unsigned H[4], L[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(H[j]) : "v"(v[2 * j]), "v"(v[2 * j + 1]));
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v[2 * j]), "v"(H[j]));
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v[2 * j + 1]), "v"(H[j]));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(L[j]) : "v"(r0), "v"(r1));
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 hv = {H[0], H[1], H[2], H[3]}, lv = {L[0], L[1], L[2], L[3]};
    hi = __builtin_bit_cast(h16x8, hv);
    lo = __builtin_bit_cast(h16x8, lv);
}

template <int NA, int KB>
__device__ __forceinline__ void dw_body_x2(const GemmDesc &g, int slice, int64_t t0, int64_t t1, float *__restrict__ partial,
                                           char *lds, int lane, int wave, float gscale) {
    constexpr int AW = 128 * NA, XW = 32 * KB;
    constexpr int A_PIECES = 32 * AW * 4 / 1024 / 4, X_PIECES = 32 * XW * 4 / 1024 / 4;   // 1-KiB DMA pieces per wave
    constexpr int A_BYTES = 32 * AW * 4, X_BYTES = 32 * XW * 4, STAGE_BYTES = A_BYTES + X_BYTES;
    constexpr int NSTAGE = (KB <= 2) ? 3 : 2;
    // KB = 1 (the 128 x 32 direction item: six MFMAs per tile) issues the next tile's five pieces in one go behind the
    // barrier.  With ALL five spread between the k-steps like the wider shapes' this one shape came back with wrong fc_9
    // direction columns on gfx950 -- deterministically, the same wrong values with two and three stages, with vmcnt(0), with a
    // pause behind the barrier and with the MFMAs drained by s_nops.  Right again: any ONE piece issued behind the last MFMAs
    // instead (or all in front), and the five landing in an unused LDS region with the real ones issued behind; a compiler
    // barrier or the M0 write alone at the same places change nothing; one tile per workgroup (the pieces then land in a stage
    // nobody reads) fails the same way; with every LDS read of the tile waited for before the first piece it is right: the ds_reads
    // behind the pieces are what comes back wrong.  The ISA's addresses, waits and operands check out line by line; why was not found (scripts/diag_f16x2_dw.py caught it; tests/test_gpu_f16x2.py compares every entry).
    constexpr bool FRONT = KB == 1;
    constexpr int PER_WAVE = A_PIECES + X_PIECES, SLOTS = 2 * KB, PPS = (PER_WAVE + SLOTS - 1) / SLOTS;
    static_assert(NSTAGE * STAGE_BYTES <= DW_LDS_BYTES, "stage ring exceeds the LDS allocation");
    static_assert((NSTAGE - 2) * PER_WAVE < 64, "counted wait");
    const int i = lane & 31, kg = lane >> 5;
    const bool want_bias = (g.flags & FLAG_BIAS) != 0;
    const char *a_src = g.a_src, *x_src = g.x_src;
    const int64_t a_stride = g.a_stride, x_stride = g.x_stride;
    const unsigned lane_off = (unsigned)lane * 16u;
    float *out = partial + g.partial_off + slice * slice_stride(g);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    // feature 32 blk + i of sample s inside a TF tile: byte blk * 4096 + 1024 q + ((32 s + 16 hh) ^ (32 q)) + 4 e with
    // q = (i >> 3) & 3, hh = (i >> 2) & 1, e = i & 3; this lane's samples of k-step ks are 16 ks + 8 kg + j, j = 0..7
    const unsigned q = (unsigned)(i >> 3) & 3u, hh = (unsigned)(i >> 2) & 1u, e = (unsigned)i & 3u;
    unsigned off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) off[j] = 1024u * q + 256u * (unsigned)kg + 16u * hh + 4u * e + 32u * ((unsigned)j ^ q);

    f32x16 acc[NA][KB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bsum[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) bsum[a] = 0.0f;

    auto issue_piece = [&](int64_t t, int buf, int j) {
        const unsigned ad = lds_base + buf * STAGE_BYTES;
        if (j < A_PIECES) lds_dma_16s(a_src + t * a_stride + (wave + 4 * j) * 1024, lane_off, ad + (wave + 4 * j) * 1024);
        else {
            const int jx = j - A_PIECES;
            lds_dma_16s(x_src + t * x_stride + (wave + 4 * jx) * 1024, lane_off, ad + A_BYTES + (wave + 4 * jx) * 1024);
        }
    };
    if (t0 < t1)
        for (int d = 0; d < NSTAGE - 1; ++d) {
            const int64_t tt = t0 + d < t1 ? t0 + d : t1 - 1;
#pragma unroll
            for (int j = 0; j < PER_WAVE; ++j) issue_piece(tt, d, j);
        }
    int buf = 0;
    for (int64_t t = t0; t < t1; ++t) {
        wait_vmcnt<(NSTAGE - 2) * PER_WAVE>();   // tile t landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();            // ... and everybody else's; stage (t-1) is free again
        asm volatile("" ::: "memory");
        const int64_t tn_raw = t + NSTAGE - 1;
        const int64_t tn = tn_raw < t1 ? tn_raw : t1 - 1;
        int nbuf = buf + NSTAGE - 1;
        if (nbuf >= NSTAGE) nbuf -= NSTAGE;
        const char *stage = lds + buf * STAGE_BYTES;
        if (FRONT) {
#pragma unroll
            for (int j = 0; j < PER_WAVE; ++j) issue_piece(tn, nbuf, j);
        }
        // Two k-steps x KB column blocks = 2 KB steps, software-pipelined by hand: step s requests the X fragment of step
        // s + 2, issues its own MFMAs and splits step s + 1's fragment in their shadow (a pair of samples = four vector
        // instructions behind each MFMA; sched_barrier pins the order -- left to itself the compiler reads, splits and only
        // then multiplies, and the matrix pipe waits out every LDS round trip: 12.4 k clocks per 256 x 256 tile; one step
        // ahead 6.3 k, two steps ahead 5.6 k; the pipe needs 3.1 k, the tile's DMA round trip 4.8 k).
        constexpr int STEPS = 2 * KB;
        auto read_x = [&](int st, float (&r)[8]) {
            const int ks = st / KB, kb = st % KB;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                r[jj] = *reinterpret_cast<const float *>(stage + A_BYTES + kb * 4096 + 512 * ks + off[jj]);
        };
        h16x8 ahi[NA], alo[NA];
        float ar[NA][8];                      // raw dY fragments of the second k-step, read at the top of the tile
        auto read_a = [&](int ks) {
#pragma unroll
            for (int nb = 0; nb < NA; ++nb)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    ar[nb][jj] = *reinterpret_cast<const float *>(stage + (wave * NA + nb) * 4096 + 512 * ks + off[jj]);
        };
        auto split_a = [&]() {
#pragma unroll
            for (int nb = 0; nb < NA; ++nb) {
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = ar[nb][jj];
                bsum[nb] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] *= gscale;
                split8(v, ahi[nb], alo[nb]);
            }
        };
        // X fragments are read TWO steps ahead (a read issued one step ahead comes back ~100 clocks after the MFMAs that
        // should hide it have drained: 6.3 k clocks per 256 x 256 tile; the matrix pipe needs 3.1 k)
        float xr[2][8];
        read_x(0, xr[0]);
        read_a(0);
        if (STEPS > 1) read_x(1, xr[1]);
        split_a();
        read_a(1);
        h16x8 bhi, blo;
        split8(xr[0], bhi, blo);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int kb = st % KB;
            const float (&cur)[8] = xr[(st + 1) & 1];       // step st + 1's fragment, requested at the top of step st - 1
            if (st + 2 < STEPS) read_x(st + 2, xr[st & 1]);  // (step st's own was split during step st - 1: its registers are free)
            unsigned H[4] = {0, 0, 0, 0}, L[4] = {0, 0, 0, 0};
            auto pair = [&](int pp) {
                if (st + 1 >= STEPS) return;
                float r0, r1;
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(H[pp]) : "v"(cur[2 * pp]), "v"(cur[2 * pp + 1]));
                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(cur[2 * pp]), "v"(H[pp]));
                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(cur[2 * pp + 1]), "v"(H[pp]));
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(L[pp]) : "v"(r0), "v"(r1));
            };
#pragma unroll
            for (int p = 0; p < PPS; ++p)
                if (!FRONT && st * PPS + p < PER_WAVE) issue_piece(tn, nbuf, st * PPS + p);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int nb = 0; nb < NA; ++nb) {
                    acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(m == 0 ? alo[nb] : ahi[nb], m == 1 ? blo : bhi, acc[nb][kb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const int idx = m * NA + nb;       // a pair of samples (four vector instructions) behind each MFMA
                    if (NA == 2 && idx <= 3) pair(idx);
                    if (NA == 1 && idx <= 1) { pair(2 * idx); pair(2 * idx + 1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            if (st + 1 == KB) split_a();      // (the MFMAs above were the last to read the first k-step's dY fragments)
            if (st + 1 < STEPS) {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                const u4 hv = {H[0], H[1], H[2], H[3]}, lv = {L[0], L[1], L[2], L[3]};
                bhi = __builtin_bit_cast(h16x8, hv);
                blo = __builtin_bit_cast(h16x8, lv);
            }
        }
        buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
    }
    // partial tile of this slice, the layout of dw_body: row-major [AW][XW], then bias[AW]
    // (the accumulators go out as they are, 2^t too large: the reducer takes the scale out of the slices' sum, which keeps
    // this flush a run of stores straight from the accumulation registers)
    {
        const unsigned lane_bytes = (unsigned)(4 * kg * XW + i) * 4u;
#pragma unroll
        for (int nb = 0; nb < NA; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *row = out + (int64_t)(wave * 32 * NA + 32 * nb + (r & 3) + 8 * (r >> 2)) * XW;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(row) + lane_bytes + kb * 128) = acc[nb][kb][r];
            }
    }
#pragma unroll
    for (int nb = 0; nb < NA; ++nb) {
        const float both = bsum[nb] + __shfl_xor(bsum[nb], 32, WAVE);
        if (want_bias && kg == 0) out[AW * XW + wave * 32 * NA + 32 * nb + i] = both;
    }
}

// 2^t that takes a gradient plane's largest |dY| into [2^9, 2^10) (1 for an all-zero or non-finite plane)
__device__ __forceinline__ float plane_scale(const unsigned *__restrict__ plane_max, int plane, bool inverse) {
    const float mx = __builtin_bit_cast(float, plane_max[plane]);
    int ex;
    (void)frexpf(mx, &ex);
    int tt = 10 - ex;
    tt = tt > 100 ? 100 : (tt < -100 ? -100 : tt);
    return (mx > 0.0f && mx < INFINITY) ? ldexpf(1.0f, inverse ? -tt : tt) : 1.0f;
}

__global__ __launch_bounds__(256, 1) void mlp_bwd_dw_x2_kernel(GemmTable table, float *__restrict__ partial, int64_t M,
                                                                const unsigned *__restrict__ plane_max, ItemPlanes planes,
                                                                unsigned long long *__restrict__ block_clocks) {
    const unsigned long long clk0 = block_clocks ? wall_clock64() : 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t MP = padded_rows(M), tiles = MP / 32;
    const int64_t B = gridDim.x, b = blockIdx.x;
    const int64_t lo = table.work_total * b / B, hi = table.work_total * (b + 1) / B;
    bool first = true;
    for (int k = 0; k < table.n; ++k) {
        const GemmDesc &g = table.g[k];
        const int slice = (int)b - g.first_block;
        if (slice < 0 || slice >= g.num_slices) continue;
        auto tile_at = [&](int64_t unit) {
            const int64_t rel = unit - g.unit_off;
            const int64_t j = rel <= 0 ? 0 : (rel + g.cost - 1) / g.cost;
            return j < tiles ? j : tiles;
        };
        const int64_t t0 = tile_at(lo), t1 = tile_at(hi);
        if (!first) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        first = false;
        const float gscale = plane_scale(plane_max, planes.p[k], false);
        if (g.a_width == 256 && g.x_width == 256) dw_body_x2<2, 8>(g, slice, t0, t1, partial, lds, lane, wave, gscale);
        else if (g.a_width == 256 && g.x_width == 64) dw_body_x2<2, 2>(g, slice, t0, t1, partial, lds, lane, wave, gscale);
        else if (g.a_width == 128 && g.x_width == 256) dw_body_x2<1, 8>(g, slice, t0, t1, partial, lds, lane, wave, gscale);
        else dw_body_x2<1, 1>(g, slice, t0, t1, partial, lds, lane, wave, gscale);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (block_clocks && threadIdx.x == 0) block_clocks[blockIdx.x] = wall_clock64() - clk0;  // 100 MHz ticks
}

// the two thin rows the fp32 dW kernel sums as side jobs: fc_8.weight[0, :] = sum_m dsig[m] h7[m, :] and fc_out.weight[c, :] =
// sum_m gy[m][c] h9[m, :] -- one pass over the h7 / h9 planes of the record (HBM-bound), double accumulation, fixed order
constexpr int X2_THIN_SLICES = 512;
// H7: blocks 0..7 of the h7 plane against dsig (16 double accumulators per lane); otherwise blocks 0..3 of the h9 plane against
// the three colour gradients (48).  Two instances instead of one kernel that carries the larger accumulator set for both: the
// common one needed 258 registers = ONE wavefront per SIMD, and a loop of dependent 2 us round trips at that occupancy took
// 450 us for 1.2 GB.
template <bool H7>
__global__ __launch_bounds__(64) void fused_thin_kernel(const float *__restrict__ saved, const float *__restrict__ dy, int64_t MP,
                                                        int slices, double *__restrict__ partial) {
    constexpr int NC = H7 ? 1 : 3, width = H7 ? 256 : 128;
    const int fb = blockIdx.x, lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const float *plane = saved + (H7 ? pl_h(MP, 7) : pl_h9(MP));
    const float *gy = dy + gy_plane(MP), *ds = dy + dsig_plane(MP);
    const int64_t tiles = MP / 32, per = (tiles + slices - 1) / slices;
    const int64_t t0 = blockIdx.y * per, t1 = t0 + per < tiles ? t0 + per : tiles;
    double acc[NC][16];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0;
    for (int64_t t = t0; t < t1; ++t) {
        const float *tile = plane + t * 32 * width + fb * 1024;
        const int64_t m = t * 32 + i;
        float x[16];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + qq * 256 + 4 * ((2 * i + h) ^ (2 * qq)));
            x[4 * qq] = v.x; x[4 * qq + 1] = v.y; x[4 * qq + 2] = v.z; x[4 * qq + 3] = v.w;
        }
        if (H7) {
            const double gd = (double)ds[m];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] += gd * (double)x[r];
        } else {
            const f32x4 g4 = *reinterpret_cast<const f32x4 *>(gy + 4 * m);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[0][r] += (double)g4.x * (double)x[r];
                acc[NC > 1 ? 1 : 0][r] += (double)g4.y * (double)x[r];
                acc[NC > 2 ? 2 : 0][r] += (double)g4.z * (double)x[r];
            }
        }
    }
    double *out = partial + (int64_t)blockIdx.y * (256 + 3 * 128);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double v = acc[c][r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);      // over the 32 samples of this lane half
            const int k = 32 * fb + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (i == 0) out[H7 ? k : 256 + c * 128 + k] = v;
        }
    }
}
__global__ __launch_bounds__(64) void fused_thin_reduce_kernel(const double *__restrict__ partial, int slices, float *__restrict__ g_params,
                                                               int64_t off_w8, int64_t off_wout) {
    const int col = blockIdx.x;     // 0..255 density row, 256..639 fc_out.weight (3 x 128, contiguous)
    __shared__ double part[64];
    double acc = 0.0;
    for (int z = threadIdx.x; z < slices; z += 64) acc += partial[(int64_t)z * 640 + col];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int t = 1; t < 64; ++t) acc += part[t];
    g_params[col < 256 ? off_w8 + col : off_wout + (col - 256)] = (float)acc;
}

// ------------------------------------------------------------------------------------------
// stage 3: reduce partial tiles into the flat gradient (state_dict layout), fixed order
// ------------------------------------------------------------------------------------------
template <bool SCALED>
__device__ __forceinline__ void reduce_item(const GemmDesc &g, const float *__restrict__ partial, int64_t off_wout,
                                            float *__restrict__ g_params, float tile_scale);

__global__ void mlp_bwd_reduce_list_kernel(const GemmList *__restrict__ list, const float *__restrict__ partial,
                                           float *__restrict__ g_params) {
    const GemmDesc *items = reinterpret_cast<const GemmDesc *>(list + 1);
    if ((int)blockIdx.y < list->n) reduce_item<false>(items[blockIdx.y], partial, list->off_wout, g_params, 1.0f);
}

template <bool SCALED>
__device__ __forceinline__ void reduce_main(const GemmTable &table, const float *__restrict__ partial,
                                            const float *__restrict__ bias_partial, int bias_partials,
                                            float *__restrict__ g_params, const unsigned *__restrict__ plane_max,
                                            const ItemPlanes *planes) {
    const int gi = blockIdx.y;
    if (gi >= table.n) {  // the four scalar-output bias gradients: per-wavefront partials of the dX chain, fixed order
        __shared__ float part[64][4];
        if (blockIdx.x != 0) return;
        const int c = threadIdx.x & 3, grp = threadIdx.x >> 2;   // 64 groups walk the partials with stride 64
        float s0 = 0.0f;
        for (int j = grp; j < bias_partials; j += 64) s0 += bias_partial[(int64_t)j * 4 + c];
        part[grp][c] = s0;
        __syncthreads();
        if (threadIdx.x < 4) {
            float t = 0.0f;
            for (int q = 0; q < 64; ++q) t += part[q][threadIdx.x];
            g_params[threadIdx.x < 3 ? table.off_bout + threadIdx.x : table.off_b8] = t;   // fc_out.bias[0..2], fc_8.bias[0]
        }
        return;
    }
    reduce_item<SCALED>(table.g[gi], partial, table.off_wout, g_params,
                        SCALED ? plane_scale(plane_max, planes->p[gi], true) : 1.0f);
}

__global__ void mlp_bwd_reduce_kernel(GemmTable table, const float *__restrict__ partial,
                                      const float *__restrict__ bias_partial, int bias_partials,
                                      float *__restrict__ g_params) {
    reduce_main<false>(table, partial, bias_partial, bias_partials, g_params, nullptr, nullptr);
}

// after mlp_bwd_dw_x2_kernel: every item's weight entries come back out of its gradient plane's power-of-two scale
__global__ void mlp_bwd_reduce_x2_kernel(GemmTable table, const float *__restrict__ partial,
                                         const float *__restrict__ bias_partial, int bias_partials,
                                         float *__restrict__ g_params, const unsigned *__restrict__ plane_max,
                                         ItemPlanes planes) {
    reduce_main<true>(table, partial, bias_partial, bias_partials, g_params, plane_max, &planes);
}

// SCALED (the split-f16 dW kernel's partial tiles): the weight entries leave multiplied by tile_scale, a power of two
template <bool SCALED>
__device__ __forceinline__ void reduce_item(const GemmDesc &g, const float *__restrict__ partial, int64_t off_wout,
                                            float *__restrict__ g_params, float tile_scale) {
    const int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    const float *base = partial + g.partial_off;
    const int64_t stride = slice_stride(g);
    const int64_t tile = (int64_t)g.valid_rows * g.valid_cols;
    const int64_t with_bias = tile + ((g.flags & FLAG_BIAS) ? g.valid_rows : 0);
    const int64_t total = with_bias + ((g.flags & FLAG_DENSITY) ? FEAT : (g.flags & FLAG_FCOUT) ? 3 * HALF : 0);
    for (int64_t e = e0; e < total; e += step) {
        int64_t src, dst;
        if (e < tile) {
            const int n = (int)(e / g.valid_cols), k = (int)(e % g.valid_cols);
            src = (int64_t)n * g.x_width + k;
            dst = g.w_off + (int64_t)(n + g.row0) * g.in_features + g.col0 + k;
        } else if (e < with_bias) {
            const int n = (int)(e - tile);
            src = (int64_t)g.a_width * g.x_width + n;
            dst = g.b_off + g.row0 + n;
        } else {   // side-job rows: density row of fc_8 = weight[0, :]; fc_out.weight (3 x 128, contiguous)
            const int k = (int)(e - with_bias);
            src = (int64_t)g.a_width * g.x_width + 256 + k;
            dst = (g.flags & FLAG_DENSITY) ? g.w_off + k : off_wout + k;
        }
        // slices are added in index order (bit-reproducible); the loads of four slices are issued together so that the
        // walk is not one dependent HBM round trip per slice
        float s = 0.0f;
        if (g.a_split > 1) {   // the k-groups of a narrow item sit a_width / a_split rows apart (rows, or bias entries)
            const int64_t fold = (int64_t)(g.a_width / g.a_split) * (e < tile ? g.x_width : 1);
            int sl = 0;
            for (; sl + 2 <= g.num_slices; sl += 2) {   // (loads of two slices issued together, added in index order)
                float p[2][4];
                for (int u = 0; u < 2; ++u)
                    for (int f = 0; f < 4; ++f) p[u][f] = f < g.a_split ? base[(int64_t)(sl + u) * stride + src + f * fold] : 0.0f;
                for (int u = 0; u < 2; ++u)
                    for (int f = 0; f < g.a_split; ++f) s += p[u][f];
            }
            for (; sl < g.num_slices; ++sl)
                for (int f = 0; f < g.a_split; ++f) s += base[(int64_t)sl * stride + src + f * fold];
            g_params[dst] = s;
            continue;
        }
        int sl = 0;
        for (; sl + 4 <= g.num_slices; sl += 4) {
            const float p0 = base[(int64_t)sl * stride + src], p1 = base[(int64_t)(sl + 1) * stride + src];
            const float p2 = base[(int64_t)(sl + 2) * stride + src], p3 = base[(int64_t)(sl + 3) * stride + src];
            s += p0; s += p1; s += p2; s += p3;
        }
        for (; sl < g.num_slices; ++sl) s += base[(int64_t)sl * stride + src];
        g_params[dst] = (SCALED && e < tile) ? s * tile_scale : s;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Plan {
    GemmTable table;
    int total_blocks;
    int64_t partial_floats;
};

// mlp_bwd_dw_x2_kernel's relative time of one 32-row tile per item shape (scripts/dw_timing.py --f16x2, MI355X, all CUs
// busy; same unit as the fp32 table in make_plan): its 256 x 256 tiles stream at 4.9 TB/s, so a tile's time follows its
// BYTES (64 / 40 / 48 / 20 KiB) far more than its MFMAs -- planned with the fp32 kernel's costs the thin items' workgroups
// finished at 5.7 ms, the wide ones' at 3.3 ms
constexpr int X2_COST_256_256 = 7350, X2_COST_256_64 = 3207, X2_COST_128_256 = 5120, X2_COST_128_32 = 1865;
Plan make_plan(const Net &net, int64_t M, int cus, const float *saved, const float *dy, bool x2 = false) {
    const int E_POS = net.e_pos, E_DIR = net.e_dir;
    const int64_t MP = padded_rows(M);
    const int64_t tiles = MP / 32;
    Plan p;
    GemmTable &T = p.table;
    int n = 0;
    auto add = [&](int layer, int64_t a_off, int a_width, int64_t x_off, int x_width, int col0, int valid_cols,
                   int row0, int flags) {
        GemmDesc &g = T.g[n++];
        g.a_src = reinterpret_cast<const char *>(dy + a_off); g.x_src = reinterpret_cast<const char *>(saved + x_off);
        g.a_stride = 128 * (int64_t)a_width; g.x_stride = 128 * (int64_t)x_width; g.valid_rows = a_width; g.a_split = 1;
        g.a_width = a_width; g.x_width = x_width;
        g.flags = flags; g.in_features = net.layer_in(layer); g.col0 = col0; g.valid_cols = valid_cols; g.row0 = row0;
        g.w_off = net.w_offset(layer); g.b_off = net.b_offset(layer);
    };
    add(0, dy_plane(MP, 0), 256, pl_pe(MP), 64, 0, E_POS, 0, FLAG_BIAS);
    for (int l = 1; l <= 4; ++l) add(l, dy_plane(MP, l), 256, pl_h(MP, l - 1), 256, 0, 256, 0, FLAG_BIAS);
    add(5, dy_plane(MP, 5), 256, pl_pe(MP), 64, 0, E_POS, 0, 0);
    add(5, dy_plane(MP, 5), 256, pl_h(MP, 4), 256, E_POS, 256, 0, FLAG_BIAS);
    add(6, dy_plane(MP, 6), 256, pl_h(MP, 5), 256, 0, 256, 0, FLAG_BIAS);
    add(7, dy_plane(MP, 7), 256, pl_h(MP, 6), 256, 0, 256, 0, FLAG_BIAS);
    add(8, dy_plane(MP, 8), 256, pl_h(MP, 7), 256, 0, 256, 1, FLAG_BIAS | FLAG_DENSITY);
    add(9, dy9_plane(MP), 128, pl_y8(MP), 256, 0, 256, 0, FLAG_BIAS | FLAG_FCOUT);
    add(9, dy9_plane(MP), 128, pl_de(MP), 32, FEAT, E_DIR, 0, 0);
    T.n = n;
    T.off_b8 = net.b_offset(8); T.off_wout = net.w_offset(10); T.off_bout = net.b_offset(10);
    // Relative time of one 32-row tile per item shape (measured on MI355X with all CUs busy, ns; scripts/dw_timing.py):
    // wide tiles are MFMA-bound (256 MFMAs per wave, 6.83 us at the peak), the thin ones lean on the DMA round trip.
    // The fc_8 item carries the density row (+1.4 %).
    int64_t units = 0;
    for (int k = 0; k < n; ++k) {
        const int aw = T.g[k].a_width, xw = T.g[k].x_width;
        // (round 4, scripts/dw_timing.py with all 256 workgroups running: the three workgroups of the thin 128 x 32 item
        // finished 2.8 % after the mean and set the kernel's duration; density / fc_out items +0.4 %)
        T.g[k].cost = (T.g[k].flags & FLAG_DENSITY) ? 7480 : (T.g[k].flags & FLAG_FCOUT) ? 4145 : (aw == 256 && xw == 256) ? 7350
                    : (aw == 256 && xw == 64) ? 2010 : (aw == 128 && xw == 256) ? 3770 : 965;
        // the split-f16 kernel: no side jobs, and a tile's time follows its bytes far more than its MFMAs (same units)
        if (x2) T.g[k].cost = (aw == 256 && xw == 256) ? X2_COST_256_256 : (aw == 256 && xw == 64) ? X2_COST_256_64
                            : (aw == 128 && xw == 256) ? X2_COST_128_256 : X2_COST_128_32;
        T.g[k].unit_off = units;
        units += tiles * T.g[k].cost;
    }
    T.work_total = units;
    const int64_t B = tiles * n < cus ? tiles * n : cus;      // tiny batches: at most one workgroup per tile
    auto owner = [&](int64_t unit) {                           // workgroup whose interval [W b / B, W (b+1) / B) holds `unit`
        int64_t b = unit * B / units;
        while (b + 1 < B && units * (b + 1) / B <= unit) ++b;
        while (b > 0 && units * b / B > unit) --b;
        return (int)b;
    };
    int64_t off = 0;
    for (int k = 0; k < n; ++k) {
        const int fb = owner(T.g[k].unit_off);
        const int lb = owner(T.g[k].unit_off + (tiles - 1) * T.g[k].cost);
        T.g[k].first_block = fb;
        T.g[k].num_slices = lb - fb + 1;
        T.g[k].partial_off = off;
        off += (int64_t)T.g[k].num_slices * ((int64_t)T.g[k].a_width * T.g[k].x_width + SLICE_EXTRA);
    }
    const int block = (int)B;
    p.total_blocks = block;
    p.partial_floats = off;
    return p;
}

inline int64_t align256f(int64_t floats) { return (floats + 63) & ~(int64_t)63; }

}  // namespace

// ------------------------------------------------------------------------------------------
// the dW GEMMs over arbitrary tile-fragment planes (mlp_layered.hip): windows of <= 256 x 256
// ------------------------------------------------------------------------------------------
namespace nerf {
struct DwItem {   // (mirrors the declaration in mlp_layered.hip)
    const float *a_plane; int a_width, a_fb0, a_blocks;
    const float *x_plane; int x_width, x_fb0, x_blocks;
    float *w_dst; int ld;
    int rows_valid, cols_valid;
    float *b_dst;
    int row0;      // rows of the destination in front of the window's first row that w_dst / b_dst do NOT include (1 for an
                   // fc_8 item with the density side job: its w_dst is the tensor's start, where the side job's row goes)
    int side;      // 0 | 2 (FLAG_DENSITY) | 4 (FLAG_FCOUT)
};
struct DwSide {    // planes of the side jobs + where fc_out.weight sits in the flat gradient (mirrors mlp_layered.hip)
    const float *h9, *dsig, *gy;
    float *wout;
};

int64_t dw_items_scratch_bytes(int n_items) {
    const int64_t list = ((int64_t)sizeof(GemmList) + (int64_t)n_items * (int64_t)sizeof(GemmDesc) + 255) & ~(int64_t)255;
    return list + 4 * (int64_t)(512 + n_items) * (256 * 256 + SLICE_EXTRA);
}

// The descriptor list travels through a small ring of PINNED host buffers: a copy from pageable memory makes the runtime
// stage the bytes synchronously inside hipMemcpyAsync (host blocked, not capturable into a graph, and correct only
// because of that staging).  A slot is reused after the event recorded behind its copy has completed.
namespace {
struct ListRing {
    static constexpr int SLOTS = 8;
    char *buf[SLOTS] = {};
    size_t cap[SLOTS] = {};
    hipEvent_t done[SLOTS] = {};
    bool used[SLOTS] = {};
    int next = 0;
    std::mutex mu;
    // -> a pinned buffer of >= bytes whose previous upload (if any) has finished; *slot for mark()
    char *take(size_t bytes, int *slot) {
        std::lock_guard<std::mutex> lock(mu);
        const int k = next;
        next = (next + 1) % SLOTS;
        if (used[k] && hipEventSynchronize(done[k]) != hipSuccess) return nullptr;
        if (cap[k] < bytes) {
            if (buf[k]) (void)hipHostFree(buf[k]);
            buf[k] = nullptr; cap[k] = 0;
            const size_t want = (bytes + 65535) & ~(size_t)65535;
            if (hipHostMalloc(reinterpret_cast<void **>(&buf[k]), want, hipHostMallocDefault) != hipSuccess) return nullptr;
            cap[k] = want;
        }
        if (!done[k] && hipEventCreateWithFlags(&done[k], hipEventDisableTiming) != hipSuccess) return nullptr;
        *slot = k;
        return buf[k];
    }
    void mark(int slot, hipStream_t s) {
        std::lock_guard<std::mutex> lock(mu);
        used[slot] = hipEventRecord(done[slot], s) == hipSuccess;
        // (an event that cannot be recorded on this stream must not leave the slot looking free while its copy is in
        // flight: drain the stream instead -- and clear the error just handled, or the check_launch behind the dW launch
        // would report it as a launch failure)
        if (!used[slot]) {
            (void)hipStreamSynchronize(s);
            (void)hipGetLastError();
        }
    }
};
constexpr int MAX_RING_DEVICES = 64;
ListRing g_list_ring[MAX_RING_DEVICES];     // one per device ordinal: a slot's event is created under the device that uses it
}  // namespace

// The host-only half of run_dw_items: descriptors, the tile -> workgroup map and what the list needs of the scratch
// buffer.  No HIP call in here: nerf_mlp_layered_plan_check (mlp_layered.hip) runs it on a machine without a GPU, under
// the CPU test suite, for every network shape -- the place where a latent overflow was found in round 5.
struct DwPlan {
    std::vector<GemmDesc> G;
    float *base = nullptr;        // lowest destination pointer of the list: offsets are taken from it
    int64_t units = 0, B = 0;     // work units of the list, workgroups of the launch
    int64_t partial_floats = 0;   // partial tiles of all slices
    int64_t list_bytes = 0;       // header + descriptors, rounded to 256
    size_t host_bytes = 0;
};
// X-window width the kernel works on, in 32-feature blocks: 1 | 2 | 3 | 4 | 8.  Three blocks = the 96-wide position window
// of coord_encode_level 11..15 next to a 256-row dY window (round 6: as a four-block window it cost 3750 units per tile
// against 2850) -- other three-block windows keep the four-block shape.
static int dw_item_kb(const DwItem &it) {
    if (it.x_blocks == 3 && it.a_blocks > 4) return 3;
    return it.x_blocks > 4 ? 8 : it.x_blocks > 2 ? 4 : it.x_blocks > 1 ? 2 : 1;
}
static void plan_dw_items(const std::vector<DwItem> &items, int64_t M, int cus, DwPlan &P) {
    const int n = (int)items.size();
    const int64_t MP = mlp::padded_rows(M), tiles = MP / 32;
    P.base = items[0].w_dst;
    for (const DwItem &it : items) {
        if (it.w_dst < P.base) P.base = it.w_dst;
        if (it.b_dst && it.b_dst < P.base) P.base = it.b_dst;
    }
    P.host_bytes = sizeof(GemmList) + (size_t)n * sizeof(GemmDesc);
    P.G.resize(n);
    int64_t units = 0;
    for (int k = 0; k < n; ++k) {
        const DwItem &it = items[k];
        GemmDesc &g = P.G[k];
        const int NA = it.a_blocks > 4 ? 2 : 1;
        const int KB = dw_item_kb(it);
        g.a_width = 128 * NA; g.x_width = 32 * KB;
        g.a_split = it.a_blocks <= 1 ? 4 : it.a_blocks <= 2 ? 2 : 1;
        // (a window wider than what is left of its plane runs on into the next tile's first blocks: finite values in
        // rows / columns the reduction never reads)
        g.a_src = reinterpret_cast<const char *>(it.a_plane + (int64_t)it.a_fb0 * 1024);
        g.x_src = reinterpret_cast<const char *>(it.x_plane + (int64_t)it.x_fb0 * 1024);
        g.a_stride = 128 * (int64_t)it.a_width; g.x_stride = 128 * (int64_t)it.x_width;
        g.flags = (it.b_dst ? FLAG_BIAS : 0) | it.side;
        g.in_features = it.ld; g.col0 = 0; g.row0 = it.row0;
        g.valid_cols = it.cols_valid; g.valid_rows = it.rows_valid;
        g.w_off = it.w_dst - P.base; g.b_off = it.b_dst ? it.b_dst - P.base : 0;
        // relative tile times by shape (measured for the fused family's four shapes, mlp_backward.hip:make_plan)
        const int nk = NA * KB;
        // (round 6, scripts/dw_list_timing.py on NeRF(75, 27, 256) and NeRF(63, 33, 256): the 256 x 96 window 2850 -> 2900, the
        // 128 x 32 one 935 -> 960, the 128 x 64 one 1050 -> 1120, the fc_out side job 4145 -> 4165 -- the three workgroups of
        // the thin direction item finished 4 % / 7 % behind the mean and set the kernel's duration)
        g.cost = (it.side & FLAG_DENSITY) ? 7480 : (it.side & FLAG_FCOUT) ? 4165
                 : nk == 16 ? 7350 : (NA == 1 && KB == 8) ? 3770 : (NA == 2 && KB == 2) ? 2010 : nk == 1 ? 960
                 : (NA == 2 && KB == 3) ? 2900 : (NA == 1 && KB == 2) ? 1120 : 450 * nk + 150;
        if (g.a_split > 1) {   // MFMA time of a wave's share, or the tile's bytes at the CU's share of HBM (~11 B / clock)
            const int mfma = 450 * KB / g.a_split + 150, hbm = 5 * (128 / g.a_split + 32 * KB);
            g.cost = mfma > hbm ? mfma : hbm;
        }
        g.unit_off = units;
        units += tiles * g.cost;
    }
    P.units = units;
    const int64_t B = tiles * n < cus ? tiles * n : cus;
    P.B = B;
    auto owner = [&](int64_t unit) {
        int64_t b = unit * B / units;
        while (b + 1 < B && units * (b + 1) / B <= unit) ++b;
        while (b > 0 && units * b / B > unit) --b;
        return (int)b;
    };
    int64_t off = 0;
    for (int k = 0; k < n; ++k) {
        GemmDesc &g = P.G[k];
        const int fb = owner(g.unit_off), lb = owner(g.unit_off + (tiles - 1) * g.cost);
        g.first_block = fb; g.num_slices = lb - fb + 1; g.partial_off = off;
        off += (int64_t)g.num_slices * ((int64_t)g.a_width * g.x_width + SLICE_EXTRA);
    }
    P.partial_floats = off;
    P.list_bytes = ((int64_t)P.host_bytes + 255) & ~(int64_t)255;
}

// bytes of the scratch buffer this list needs on a device of `cus` compute units (descriptor list + partial tiles)
int64_t dw_items_needed_bytes(const std::vector<DwItem> &items, int64_t M, int cus) {
    if (items.empty() || M <= 0 || cus <= 0) return 0;
    DwPlan P;
    plan_dw_items(items, M, cus, P);
    return P.list_bytes + 4 * P.partial_floats;
}

// floats, counted from the start of its dY / X plane, up to which the dW kernel READS for this item: the last 32-sample
// tile's window of the width the kernel works on (128 | 256 / a_split dY features, 32 | 64 | 128 | 256 X features), which
// may be wider than what is left of the plane
void dw_item_read_extent(const DwItem &it, int64_t M, int64_t *a_floats, int64_t *x_floats) {
    const int64_t tiles = mlp::padded_rows(M) / 32;
    const int NA = it.a_blocks > 4 ? 2 : 1;
    const int KB = dw_item_kb(it);
    const int a_split = it.a_blocks <= 1 ? 4 : it.a_blocks <= 2 ? 2 : 1;
    *a_floats = (int64_t)it.a_fb0 * 1024 + (tiles - 1) * 32 * (int64_t)it.a_width + 32 * (int64_t)(128 * NA / a_split);
    *x_floats = (int64_t)it.x_fb0 * 1024 + (tiles - 1) * 32 * (int64_t)it.x_width + 32 * (int64_t)(32 * KB);
}

int run_dw_items(const std::vector<DwItem> &items, int64_t M, void *scratch, int64_t scratch_bytes, hipStream_t s,
                 const DwSide *side) {
    const int n = (int)items.size();
    if (n == 0 || M <= 0) return NERF_OK;
    static nerf::DeviceMask configured{0};
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_bwd_dw_list_kernel), DW_LDS_BYTES, configured,
                                          "nerf_mlp_layered_backward: LDS attribute (dW)"))
        return rc;
    DwPlan P;
    plan_dw_items(items, M, nerf::device_cus(), P);
    if (P.list_bytes + 4 * P.partial_floats > scratch_bytes)
        return nerf::fail(NERF_ERR_ARG, "nerf_mlp_layered_backward: workspace too small for the dW partial tiles");
    int slot = 0, device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= MAX_RING_DEVICES)
        return nerf::fail(NERF_ERR_LAUNCH, "nerf_mlp_layered_backward: device ordinal outside the pinned-list rings");
    ListRing &ring = g_list_ring[device];       // (events belong to the device they were created under: one ring each)
    char *host = ring.take(P.host_bytes, &slot);
    if (!host) return nerf::fail(NERF_ERR_LAUNCH, "nerf_mlp_layered_backward: pinned staging buffer for the dW item list");
    GemmList *hdr = reinterpret_cast<GemmList *>(host);
    static_assert(sizeof(GemmList) % 8 == 0, "the descriptors behind the header stay 8-byte aligned");
    memcpy(host + sizeof(GemmList), P.G.data(), (size_t)n * sizeof(GemmDesc));
    hdr->n = n; hdr->work_total = P.units;
    hdr->side_h9 = side ? side->h9 : nullptr; hdr->side_dsig = side ? side->dsig : nullptr; hdr->side_gy = side ? side->gy : nullptr;
    hdr->off_wout = side ? side->wout - P.base : 0;
    if (hipMemcpyAsync(scratch, host, P.host_bytes, hipMemcpyHostToDevice, s) != hipSuccess)
        return nerf::check_launch("nerf_mlp_layered_backward: item list upload");
    ring.mark(slot, s);
    const GemmList *list = static_cast<const GemmList *>(scratch);
    float *partial = reinterpret_cast<float *>(static_cast<char *>(scratch) + P.list_bytes);
    // NERF_DW_TIMING=<file>: debugging aid, dumps per-workgroup durations of the dW kernel (syncs!), as for the fused family
    const char *timing_path = getenv("NERF_DW_TIMING");
    unsigned long long *clocks = nullptr;
    if (timing_path && hipMalloc(&clocks, sizeof(unsigned long long) * P.B) != hipSuccess) clocks = nullptr;
    hipLaunchKernelGGL(mlp_bwd_dw_list_kernel, dim3((unsigned)P.B), dim3(256), DW_LDS_BYTES, s, list, partial, M, clocks);
    if (int rc = nerf::check_launch("nerf_mlp_layered_backward: dW")) return rc;
    if (clocks) {
        std::vector<unsigned long long> host(P.B);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(host.data(), clocks, sizeof(unsigned long long) * P.B, hipMemcpyDeviceToHost);
        (void)hipFree(clocks);
        if (FILE *f = fopen(timing_path, "w")) {
            for (int64_t b = 0; b < P.B; ++b) {
                int item = 0;
                for (int k = 0; k < n; ++k)
                    if (b >= P.G[k].first_block && b < P.G[k].first_block + P.G[k].num_slices) { item = k; break; }
                fprintf(f, "%d %d %d %lld %llu %lld %d\n", item, P.G[item].a_width, P.G[item].x_width, (long long)b, host[b],
                        (long long)P.G[item].cost, P.G[item].flags);
            }
            fclose(f);
        }
    }
    hipLaunchKernelGGL(mlp_bwd_reduce_list_kernel, dim3(64, n), dim3(256), 0, s, list,
                       static_cast<const float *>(partial), P.base);
    return nerf::check_launch("nerf_mlp_layered_backward: reduce");
}
}  // namespace nerf

NERF_API int64_t nerf_mlp_backward_workspace_bytes(const nerf_net_t *net, int64_t M) {
    mlp::Net n;
    if (nerf::fused_net(net, n, "nerf_mlp_backward_workspace_bytes") != NERF_OK) return -1;
    if (M <= 0) return 0;
    const int64_t MP = mlp::padded_rows(M);
    // upper bound on the partial buffer that does not depend on the device: 2 x 256 slices of a full tile
    const int64_t partial = (int64_t)(2 * 256 + MAX_GEMMS) * (256 * 256 + SLICE_EXTRA);
    return 4 * (align256f(MP * (int64_t)mlp::DY_FLOATS_PER_SAMPLE) + partial + (int64_t)BIAS_PARTIAL_FLOATS);
}

// floats of the workspace behind the gradient planes that a launch touches: partial tiles, bias partials, and on the split
// path the planes' maxima (16 words in a 64-float slot) and the thin rows' partial sums (doubles)
static int64_t tail_floats_needed(const Plan &plan, bool split) {
    return plan.partial_floats + BIAS_PARTIAL_FLOATS + (split ? 64 + 2 * (int64_t)X2_THIN_SLICES * 640 : 0);
}

NERF_API int nerf_mlp_backward_plan_check(const nerf_net_t *net_abi, int64_t M, int cus, int f16x2) {
    mlp::Net net;
    if (int rc = nerf::fused_net(net_abi, net, "nerf_mlp_backward_plan_check")) return rc;
    NERF_REQUIRE(M >= 0 && M <= ((int64_t)1 << 31), "nerf_mlp_backward_plan_check: M out of range");
    if (M == 0) return NERF_OK;
    if (cus <= 0) cus = 256;
    NERF_REQUIRE(!f16x2 || cus <= 384, "nerf_mlp_backward_plan_check: the split path's workspace layout assumes at most 384 compute units");
    const int64_t MP = mlp::padded_rows(M), tiles = MP / 32;
    const float *nowhere = reinterpret_cast<const float *>(uintptr_t(1) << 20);      // (the items' addresses are not looked at)
    const Plan plan = make_plan(net, M, cus, nowhere, nowhere, f16x2 != 0);
    const GemmTable &T = plan.table;
    char why[200];
    const int64_t B = plan.total_blocks;
    NERF_REQUIRE(B >= 1 && B <= cus, "nerf_mlp_backward_plan_check: workgroup count outside [1, cus]");
    int64_t partial = 0;
    for (int k = 0; k < T.n; ++k) {
        const GemmDesc &g = T.g[k];
        int64_t next = 0;      // first tile not yet covered
        for (int64_t b = 0; b < B; ++b) {
            const int64_t lo = T.work_total * b / B, hi = T.work_total * (b + 1) / B;
            auto tile_at = [&](int64_t unit) {
                const int64_t rel = unit - g.unit_off;
                const int64_t j = rel <= 0 ? 0 : (rel + g.cost - 1) / g.cost;
                return j < tiles ? j : tiles;
            };
            const int64_t t0 = tile_at(lo), t1 = tile_at(hi);
            const bool inside = b >= g.first_block && b < g.first_block + g.num_slices;
            if (t0 < t1 && (!inside || t0 != next)) {
                snprintf(why, sizeof why, "nerf_mlp_backward_plan_check: item %d, workgroup %lld: tiles [%lld, %lld) %s", k,
                         (long long)b, (long long)t0, (long long)t1, inside ? "do not continue the previous workgroup's" : "outside the item's slices");
                return nerf::fail(NERF_ERR_ARG, why);
            }
            if (t0 < t1) next = t1;
        }
        if (next != tiles) {
            snprintf(why, sizeof why, "nerf_mlp_backward_plan_check: item %d: %lld of %lld tiles covered", k, (long long)next, (long long)tiles);
            return nerf::fail(NERF_ERR_ARG, why);
        }
        NERF_REQUIRE(g.partial_off == partial, "nerf_mlp_backward_plan_check: partial tiles not packed back to back");
        partial += (int64_t)g.num_slices * ((int64_t)g.a_width * g.x_width + SLICE_EXTRA);
    }
    NERF_REQUIRE(partial == plan.partial_floats, "nerf_mlp_backward_plan_check: partial_floats does not match the items");
    const int64_t have = nerf_mlp_backward_workspace_bytes(net_abi, M) / 4 - align256f(MP * (int64_t)mlp::DY_FLOATS_PER_SAMPLE);
    if (tail_floats_needed(plan, f16x2 != 0) > have) {
        snprintf(why, sizeof why, "nerf_mlp_backward_plan_check: %lld floats behind the gradient planes, the workspace has %lld",
                 (long long)tail_floats_needed(plan, f16x2 != 0), (long long)have);
        return nerf::fail(NERF_ERR_ARG, why);
    }
    return NERF_OK;
}

namespace nerf {   // mlp_forward_f16x2.hip: stage 1 on the split-f16 kernel
int launch_dx_f16x2(const void *packed_f16x2, int64_t M, const float *sigma, const float *rgb, const float *g_sigma,
                    const float *g_rgb, const float *saved, float *dy, float *bias_partial, int *partials, unsigned *plane_max,
                    hipStream_t s);
}

// NERF_DW_TIMING=<file>: one line per workgroup -- first item it works on, its window, workgroup, clocks (100 MHz), the
// item's planning cost; syncs
static void dump_block_clocks(unsigned long long *clocks, const char *timing_path, const Plan &plan, hipStream_t s) {
    if (!clocks) return;
    std::vector<unsigned long long> host(plan.total_blocks);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(host.data(), clocks, sizeof(unsigned long long) * plan.total_blocks, hipMemcpyDeviceToHost);
    (void)hipFree(clocks);
    FILE *f = fopen(timing_path, "w");
    if (!f) return;
    for (int b = 0; b < plan.total_blocks; ++b) {
        int item = 0;
        for (int k = 0; k < plan.table.n; ++k)
            if (b >= plan.table.g[k].first_block && b < plan.table.g[k].first_block + plan.table.g[k].num_slices) { item = k; break; }
        fprintf(f, "%d %d %d %d %llu %lld\n", item, plan.table.g[item].a_width, plan.table.g[item].x_width, b, host[b],
                (long long)plan.table.g[item].cost);
    }
    fclose(f);
}

// packed_f16x2 != NULL (and no input gradients asked for): stage 1, the reverse chain, runs on the split-f16 kernel; the
// dW GEMMs run on the f16 pipe too (mlp_bwd_dw_x2_kernel: same items, work partition and partial-tile format), the two thin
// side rows in a pass of their own, and the reduction takes the gradient planes' power-of-two scales out again
static int backward_impl(const nerf_net_t *net_abi, const void *packed, const void *packed_f16x2, int64_t M, const float *sigma,
                         const float *rgb, const void *saved, const float *g_sigma, const float *g_rgb,
                         float *g_params, float *g_pos, float *g_view_dir, void *workspace,
                         nerf_stream_t stream) {
    mlp::Net net;
    if (int rc = nerf::fused_net(net_abi, net, "nerf_mlp_backward")) return rc;
    NERF_REQUIRE(M >= 0, "nerf_mlp_backward: negative M");
    NERF_REQUIRE(g_params, "nerf_mlp_backward: null g_params");
    hipStream_t s = nerf::as_stream(stream);
    if (M == 0) {   // (g_pos / g_view_dir have no rows)
        if (hipMemsetAsync(g_params, 0, sizeof(float) * net.param_count(), s) != hipSuccess)
            return nerf::check_launch("nerf_mlp_backward: memset");
        return NERF_OK;
    }
    NERF_REQUIRE(packed && sigma && rgb && saved && g_sigma && g_rgb && workspace,
                 "nerf_mlp_backward: null pointer");
    static nerf::DeviceMask configured_dx[2] = {{0}, {0}}, configured_dw{0};
    const bool input_grads = g_pos || g_view_dir;
    auto dx_kernel = input_grads ? mlp_bwd_dx_kernel<true> : mlp_bwd_dx_kernel<false>;
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(dx_kernel), mlp::LDS_BYTES,
                                          configured_dx[input_grads], "nerf_mlp_backward: LDS attribute (dX)"))
        return rc;
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_bwd_dw_kernel), DW_LDS_BYTES,
                                          configured_dw, "nerf_mlp_backward: LDS attribute (dW)"))
        return rc;
    const int cus = nerf::device_cus();
    const int64_t MP = mlp::padded_rows(M);
    float *dy = static_cast<float *>(workspace);
    float *partial = dy + align256f(MP * (int64_t)mlp::DY_FLOATS_PER_SAMPLE);
    const float *sv = static_cast<const float *>(saved);
    const bool split = packed_f16x2 && !(g_pos || g_view_dir);
    const Plan plan = make_plan(net, M, cus, sv, dy, split);
    float *bias_partial = partial + plan.partial_floats;

    const int64_t ntiles = MP / mlp::TILE_SAMPLES;
    const unsigned dx_grid = (unsigned)(ntiles < cus ? ntiles : (cus < 1024 ? cus : 1024));
    int bias_partials = (int)dx_grid * 4;
    int rc;
    // (split path: 16 words of per-plane |dY| maxima + the thin rows' partial sums live behind the bias partials -- inside the
    // workspace: its partial-tile area is sized for 2 x 256 + 13 slices, a plan uses at most cus + 12)
    unsigned *plane_max = reinterpret_cast<unsigned *>(bias_partial + BIAS_PARTIAL_FLOATS);
    double *thin_partial = reinterpret_cast<double *>(bias_partial + BIAS_PARTIAL_FLOATS + 64);
    if (split) {
        NERF_REQUIRE(cus <= 384, "nerf_mlp_backward_f16x2: workspace layout assumes at most 384 compute units");
        if (hipMemsetAsync(plane_max, 0, 64, s) != hipSuccess) return nerf::check_launch("nerf_mlp_backward_f16x2: memset");
        rc = nerf::launch_dx_f16x2(packed_f16x2, M, sigma, rgb, g_sigma, g_rgb, sv, dy, bias_partial, &bias_partials, plane_max, s);
    } else {
        hipLaunchKernelGGL(dx_kernel, dim3(dx_grid), dim3(256), mlp::LDS_BYTES, s,
                           static_cast<const char *>(packed), M, sigma, rgb, g_sigma, g_rgb, sv, dy, bias_partial);
        rc = nerf::check_launch("nerf_mlp_backward: dx chain");
    }
    if (rc != NERF_OK) return rc;
    if (input_grads) {
        const int64_t total = M * (int64_t)(net.e_pos + net.e_dir);
        hipLaunchKernelGGL(input_grad_rows_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)),
                           dim3(256), 0, s, static_cast<const float *>(dy), M, net.e_pos, net.e_dir, g_pos, g_view_dir);
        if ((rc = nerf::check_launch("nerf_mlp_backward: input gradient rows")) != NERF_OK) return rc;
    }
    // NERF_DW_TIMING=<file>: debugging aid, dumps per-workgroup durations of the dW kernel (syncs!)
    const char *timing_path = getenv("NERF_DW_TIMING");
    unsigned long long *clocks = nullptr;
    if (timing_path && hipMalloc(&clocks, sizeof(unsigned long long) * plan.total_blocks) != hipSuccess) clocks = nullptr;
    if (split) {
        static nerf::DeviceMask configured_x2{0};
        if (int rc2 = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_bwd_dw_x2_kernel), DW_LDS_BYTES, configured_x2,
                                               "nerf_mlp_backward_f16x2: LDS attribute (dW)"))
            return rc2;
        GemmTable tx = plan.table;             // same items and work partition; the two side rows go to the thin pass
        ItemPlanes planes;
        static const signed char item_plane[12] = {0, 1, 2, 3, 4, 5, 5, 6, 7, 8, 9, 9};     // make_plan's order
        for (int k = 0; k < tx.n; ++k) { tx.g[k].flags &= FLAG_BIAS; planes.p[k] = item_plane[k]; }
        hipLaunchKernelGGL(mlp_bwd_dw_x2_kernel, dim3((unsigned)plan.total_blocks), dim3(256), DW_LDS_BYTES, s, tx, partial, M,
                           static_cast<const unsigned *>(plane_max), planes, clocks);
        if ((rc = nerf::check_launch("nerf_mlp_backward_f16x2: dW")) != NERF_OK) return rc;
        dump_block_clocks(clocks, timing_path, plan, s);
        int slices = (int)(MP / 32 / 8);
        slices = slices > X2_THIN_SLICES ? X2_THIN_SLICES : (slices < 1 ? 1 : slices);
        hipLaunchKernelGGL(fused_thin_kernel<true>, dim3(8, slices), dim3(64), 0, s, sv, static_cast<const float *>(dy), MP, slices, thin_partial);
        hipLaunchKernelGGL(fused_thin_kernel<false>, dim3(4, slices), dim3(64), 0, s, sv, static_cast<const float *>(dy), MP, slices, thin_partial);
        hipLaunchKernelGGL(mlp_bwd_reduce_x2_kernel, dim3(256, tx.n + 1), dim3(256), 0, s, tx,
                           static_cast<const float *>(partial), static_cast<const float *>(bias_partial), bias_partials, g_params,
                           static_cast<const unsigned *>(plane_max), planes);
        hipLaunchKernelGGL(fused_thin_reduce_kernel, dim3(640), dim3(64), 0, s, static_cast<const double *>(thin_partial), slices,
                           g_params, (int64_t)net.w_offset(8), (int64_t)net.w_offset(10));
        return nerf::check_launch("nerf_mlp_backward_f16x2: reduce");
    }
    hipLaunchKernelGGL(mlp_bwd_dw_kernel, dim3((unsigned)plan.total_blocks), dim3(256), DW_LDS_BYTES, s,
                       plan.table, sv, static_cast<const float *>(dy), partial, M, clocks);
    rc = nerf::check_launch("nerf_mlp_backward: dW");
    if (rc != NERF_OK) return rc;
    dump_block_clocks(clocks, timing_path, plan, s);
    hipLaunchKernelGGL(mlp_bwd_reduce_kernel, dim3(256, plan.table.n + 1), dim3(256), 0, s, plan.table,
                       static_cast<const float *>(partial), static_cast<const float *>(bias_partial), bias_partials,
                       g_params);
    return nerf::check_launch("nerf_mlp_backward: reduce");
}

NERF_API int nerf_mlp_backward(const nerf_net_t *net_abi, const void *packed, const float *params, const float *pos,
                               const float *view_dir, int64_t M, int encoded, const float *sigma,
                               const float *rgb, const void *saved, const float *g_sigma, const float *g_rgb,
                               float *g_params, float *g_pos, float *g_view_dir, void *workspace,
                               nerf_stream_t stream) {
    (void)params; (void)pos; (void)view_dir; (void)encoded;  // the saved record holds the encodings
    return backward_impl(net_abi, packed, nullptr, M, sigma, rgb, saved, g_sigma, g_rgb, g_params, g_pos, g_view_dir,
                         workspace, stream);
}

// The same with the reverse chain (dX) on the split-f16 kernel: packed_f16x2 = the nerf_mlp_pack_f16x2 stream of the same
// parameters (its transposed half).  Parameter gradients only (input gradients: nerf_mlp_backward).
NERF_API int nerf_mlp_backward_f16x2(const nerf_net_t *net_abi, const void *packed, const void *packed_f16x2, int64_t M,
                                     const float *sigma, const float *rgb, const void *saved, const float *g_sigma,
                                     const float *g_rgb, float *g_params, void *workspace, nerf_stream_t stream) {
    NERF_REQUIRE(packed_f16x2, "nerf_mlp_backward_f16x2: null packed_f16x2");
    return backward_impl(net_abi, packed, packed_f16x2, M, sigma, rgb, saved, g_sigma, g_rgb, g_params, nullptr, nullptr,
                         workspace, stream);
}
