// bf16 inference variant of the fused encode + MLP kernel (BASELINE configs[2]: "bf16 MLP weights on
// MFMA").  Same algebra as mlp_forward.hip -- Y^T = W X^T, the D fragment of one layer is the B
// fragment of the next -- on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate):
//   * weights AND layer inputs are bf16 (round-to-nearest-even), accumulation, bias, ReLU, the
//     density row of fc_8, fc_out and the sigmoid stay fp32
//   * one wavefront owns 32 samples (NCB column blocks of 32; NCB = 2 is written but spills):
//     every A fragment is one ds_read_b128 = 8 bf16 out of a conflict-free swizzled image
//   * the weight stream moves in 64-KiB steps (half a layer) through a 2-step LDS ring; the next
//     step's 16 DMA instructions per wave are spread between the MFMAs of the current one
//   * positional encodings use one accurate sincos per channel and the double-angle recurrence
//     for the higher octaves (error <= 2^9 x 1e-7 = 5e-5, far below bf16 resolution 4e-3); they are
//     recomputed where needed (fc_in, fc_5, fc_9) instead of being held in registers
// Parity target: PSNR against the fp32 path (tests/test_gpu_bf16.py), not the 1e-5 bound.
#include "mlp_device.h"

namespace {

using namespace mlp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// column blocks (of 32 samples) per wavefront.  2 would halve LDS reads and the weight stream per
// sample, but 256 accumulators + 128 activation registers do not fit without spilling (hipcc 7.2).
// Also tried: 8 wavefronts per workgroup (two per SIMD, 256 registers each, one wave's vector work
// under the other's MFMAs): hipcc's schedule of the unrolled step needs ~550 registers and spills
// ~300 under that budget, whatever sched_barrier placement -- it needs a hand-scheduled step.
constexpr int NCB = 1;
constexpr int TILE = 128 * NCB;  // samples per workgroup pass

struct StepPipe {
    const char *src_wave;  // stream base + wave * 16 KiB (wave-uniform)
    unsigned lane_off;     // lane * 16
    unsigned lds_wave;     // LDS address of ring slot 0 + wave * 16 KiB
    unsigned issued;
    int issue_pos;
    unsigned consumed;

    // piece p (0..15) of the next step: this wave copies a contiguous 16 KiB quarter of the step
    __device__ __forceinline__ void issue_piece(int p) const {
        lds_dma_16s(src_wave + (size_t)issue_pos * B16_STEP_BYTES + p * 1024, lane_off,
                    lds_wave + (issued & 1) * B16_STEP_BYTES + p * 1024);
    }
    __device__ __forceinline__ void issue_done() {
        ++issued;
        issue_pos = (issue_pos + 1 == B16_STEPS) ? 0 : issue_pos + 1;
    }
    __device__ __forceinline__ unsigned acquire() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned off = (consumed & 1) * B16_STEP_BYTES;
        ++consumed;
        return off;
    }
};

// acc[cb][fb] += W[32 fb.., 32 k of this chunk] . B   for every column block; b[cb][s] are the two
// k-steps (16 features each) of the 32-feature input block.  N_PIECES > 0: also issue DMA pieces
// 0 .. N_PIECES-1 of the next step, evenly spread over the (s, fb) groups of this chunk.
__device__ __forceinline__ bf16x8 lds_read_fragment16(unsigned lds_addr, int imm_offset) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(imm_offset));
    return v;
}

template <int NFB, int N_PIECES>
__device__ __forceinline__ void mma_chunk16(f32x16 (&acc)[NCB][8], const bf16x8 (&b)[NCB][2], const char *chunk,
                                            const int (&offs)[2], const StepPipe &pipe) {
    constexpr int GROUPS = 2 * NFB, EVERY = N_PIECES > 0 ? GROUPS / N_PIECES : 1;
    static_assert(N_PIECES == 0 || GROUPS % N_PIECES == 0, "pieces must divide the groups");
    // A fragments two groups ahead in three rotating buffers (a bf16 MFMA lasts only 32 cycles, an LDS
    // read ~100): hand-issued reads, see lds_read_fragment in mlp_device.h
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)chunk;
    const unsigned addr[2] = {base + (unsigned)offs[0], base + (unsigned)offs[1]};
    bf16x8 abuf[4];
    abuf[0] = lds_read_fragment16(addr[0], 0);
    abuf[1] = lds_read_fragment16(addr[1 / NFB], (1 % NFB) * 2048);
    abuf[2] = lds_read_fragment16(addr[2 / NFB], (2 % NFB) * 2048);
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const int s = g / NFB, fb = g % NFB;
        // wait until at most the two younger reads are outstanding: fragment g has landed
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < GROUPS) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (g + 1 < GROUPS) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (g + 3 < GROUPS) abuf[(g + 3) & 3] = lds_read_fragment16(addr[(g + 3) / NFB], ((g + 3) % NFB) * 2048);
        const bf16x8 a = abuf[g & 3];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            acc[cb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[cb][s], acc[cb][fb], 0, 0, 0);
        if (N_PIECES > 0 && g % EVERY == 0) pipe.issue_piece(g / EVERY);
    }
}

// 16 fp32 values of one D-fragment block -> the two bf16 B fragments (k-steps) of the next layer
__device__ __forceinline__ void pack_block(const f32x16 &x, bf16x8 (&frag)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) frag[s][e] = (__bf16)x[8 * s + e];
}

// all NF (64 | 32) encoding features of one sample (the tail beyond 3 + 6 LEVELS is zero),
// double-angle recurrence per channel
template <int LEVELS, int NF>
__device__ __forceinline__ void encode_all(float x, float y, float z, float (&F)[NF]) {
#pragma unroll
    for (int k = 0; k < NF; ++k) F[k] = 0.0f;
    const float v[3] = {x, y, z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        F[c] = v[c];
        float s, co;
        sincos_cw<false>(v[c], s, co);  // |coordinate| itself is the only argument: no large-range issue below 3e4
#pragma unroll
        for (int f = 0; f < LEVELS; ++f) {
            F[3 + 6 * f + c] = s;
            F[3 + 6 * f + 3 + c] = co;
            const float s2 = 2.0f * s * co;
            co = fmaf(-2.0f * s, s, 1.0f);
            s = s2;
        }
    }
}

// B fragments of encoding block `blk` (features 32 blk ..) for this lane half
template <int NF>
__device__ __forceinline__ void encoding_frags(const float (&F)[NF], int blk, int h, bf16x8 (&frag)[2]) {
    f32x16 x;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = 32 * blk + (r & 3) + 8 * (r >> 2);
        x[r] = h ? F[k + 4] : F[k];
    }
    pack_block(x, frag);
}

__global__ __launch_bounds__(256, 1) void mlp_forward_bf16_kernel(const char *__restrict__ packed,
                                                                   const float *__restrict__ pos,
                                                                   const float *__restrict__ dir, int64_t M,
                                                                   float *__restrict__ sigma_out,
                                                                   float *__restrict__ rgb_out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float *cb_ = reinterpret_cast<float *>(lds + 2 * B16_STEP_BYTES);
    for (int e = tid; e < CONST_FLOATS / 4; e += 256)
        reinterpret_cast<f32x4 *>(cb_)[e] = reinterpret_cast<const f32x4 *>(packed)[e];

    int offs[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) offs[s] = b16_frag_offset(i, 2 * s + h);

    StepPipe pipe;
    pipe.src_wave = packed + CONST_BYTES + wave * 16384;
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 16384u;
    pipe.issued = 0;
    pipe.issue_pos = 0;
    pipe.consumed = 0;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; ++p) pipe.issue_piece(p);
    pipe.issue_done();

    const int64_t ntiles = (M + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int64_t m[NCB];
        bool valid[NCB];
        float raw[NCB][6];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            m[cb] = tile * TILE + wave * 32 * NCB + cb * 32 + i;
            valid[cb] = m[cb] < M;
            const int64_t mc = valid[cb] ? m[cb] : M - 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                raw[cb][c] = pos[3 * mc + c];
                raw[cb][3 + c] = dir[3 * mc + c];
            }
        }

        f32x16 acc[NCB][8];      // [column block][output feature block]
        bf16x8 act[8][NCB][2];   // [32-feature input block][column block][k-step]
        float sigma_pre[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) sigma_pre[cb] = 0.0f;

        // encoded position as B fragments, evaluated on demand (fc_in and the fc_5 skip connection)
        auto position_frags = [&](bf16x8 (&pe)[2][NCB][2]) {  // [blk][cb][s]
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                float F[64];
                encode_all<L_POS, 64>(raw[cb][0], raw[cb][1], raw[cb][2], F);
                encoding_frags(F, 0, h, pe[0][cb]);
                encoding_frags(F, 1, h, pe[1][cb]);
            }
        };
        // a 64-KiB step of which only the first two chunks are multiplied (the encoded position)
        auto position_step = [&](const bf16x8 (&pe)[2][NCB][2]) {
            const char *w = lds + pipe.acquire();
            // the whole next step is requested during the first chunk: its youngest piece still has
            // three chunks' worth of MFMAs to land before the next acquire
            mma_chunk16<8, 16>(acc, pe[0], w, offs, pipe);
            mma_chunk16<8, 0>(acc, pe[1], w + B16_CHUNK_BYTES, offs, pipe);
            pipe.issue_done();
        };
        // activation of a finished 256-wide layer -> packed bf16 inputs of the next one; then the
        // accumulators restart from the next layer's bias (the C fragment starts as the bias)
        auto finish_layer = [&](bool relu, bool density, const float *next_bias, bool next_half) {
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    f32x16 x;
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r] = relu ? relu1(acc[cb][fb][r]) : acc[cb][fb][r];
                    if (density) sigma_pre[cb] += block_dot(cb_ + CB_W8ROW0 + 32 * fb, x, h);
                    pack_block(x, act[fb][cb]);
                }
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                if (next_half) load_bias<4>(acc[cb], next_bias, h);
                else load_bias<8>(acc[cb], next_bias, h);
            }
        };

        // ---- fc_in (nerf.py:102): step 0
        {
            bf16x8 pe[2][NCB][2];
            position_frags(pe);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) load_bias<8>(acc[cb], cb_ + CB_BIAS, h);
            position_step(pe);
        }

        // ---- fc_1 .. fc_8 (nerf.py:103-113); skip connection at fc_5 (pos FIRST, :108)
        for (int l = 1; l <= 8; ++l) {
            if (l == 5) {
                // encode first (64 temporaries) while layer 4's outputs still sit in the accumulators,
                // then finish layer 4 into `act`: keeps the VGPR peak at act + pe
                bf16x8 pe[2][NCB][2];
                position_frags(pe);
                finish_layer(true, false, cb_ + CB_BIAS + 5 * 256, false);
                position_step(pe);
            } else {
                // ReLU(layer l-1); h7 feeds the density row; restart from bias_l (fc_8: rows 1..256)
                finish_layer(true, l == 8, l < 8 ? cb_ + CB_BIAS + l * 256 : cb_ + CB_BIAS8, false);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const char *w = lds + pipe.acquire();
                mma_chunk16<8, 16>(acc, act[4 * st], w, offs, pipe);
                mma_chunk16<8, 0>(acc, act[4 * st + 1], w + B16_CHUNK_BYTES, offs, pipe);
                mma_chunk16<8, 0>(acc, act[4 * st + 2], w + 2 * B16_CHUNK_BYTES, offs, pipe);
                mma_chunk16<8, 0>(acc, act[4 * st + 3], w + 3 * B16_CHUNK_BYTES, offs, pipe);
                pipe.issue_done();
            }
        }

        // ---- fc_9 on cat([x[:,1:], view_dir]) (:116-118): fc_8 has no ReLU (:113)
        finish_layer(false, false, cb_ + CB_BIAS9, true);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const char *w = lds + pipe.acquire();
            mma_chunk16<4, 8>(acc, act[4 * st], w, offs, pipe);
            {   // pieces 8..15 ride on the second chunk
                StepPipe second = pipe;
                second.src_wave += 8 * 1024;
                second.lds_wave += 8 * 1024;
                mma_chunk16<4, 8>(acc, act[4 * st + 1], w + B16_CHUNK_BYTES, offs, second);
            }
            mma_chunk16<4, 0>(acc, act[4 * st + 2], w + 2 * B16_CHUNK_BYTES, offs, pipe);
            mma_chunk16<4, 0>(acc, act[4 * st + 3], w + 3 * B16_CHUNK_BYTES, offs, pipe);
            pipe.issue_done();
        }
        {
            bf16x8 de[NCB][2];  // [cb][s]
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                float F[32];
                encode_all<L_DIR, 32>(raw[cb][3], raw[cb][4], raw[cb][5], F);
                encoding_frags(F, 0, h, de[cb]);
            }
            const char *w = lds + pipe.acquire();
            mma_chunk16<4, 8>(acc, de, w, offs, pipe);
#pragma unroll
            for (int p = 8; p < 16; ++p) pipe.issue_piece(p);
            pipe.issue_done();
        }

        // ---- ReLU(fc_9), fc_out, sigmoid (:118-119) and sigma = relu(x[:,0]) (:115), fp32 vector ALU
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float y[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                f32x16 x;
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = relu1(acc[cb][fb][r]);
#pragma unroll
                for (int c = 0; c < 3; ++c) y[c] += block_dot(cb_ + CB_WOUT + c * HALF + 32 * fb, x, h);
            }
            const float sp = sigma_pre[cb] + __shfl_xor(sigma_pre[cb], 32, WAVE);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float p = y[c] + __shfl_xor(y[c], 32, WAVE);
                y[c] = 1.0f / (1.0f + expf(-(p + cb_[CB_SCALARS + 1 + c])));
            }
            if (valid[cb] && h == 0) {
                sigma_out[m[cb]] = fmaxf(sp + cb_[CB_SCALARS], 0.0f);
                rgb_out[3 * m[cb] + 0] = y[0];
                rgb_out[3 * m[cb] + 1] = y[1];
                rgb_out[3 * m[cb] + 2] = y[2];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

NERF_API int nerf_mlp_forward_bf16(const void *packed_bf16, const float *pos, const float *view_dir, int64_t M,
                                   float *sigma, float *rgb, nerf_stream_t stream) {
    NERF_REQUIRE(M >= 0, "nerf_mlp_forward_bf16: negative M");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(packed_bf16 && pos && view_dir && sigma && rgb, "nerf_mlp_forward_bf16: null pointer");
    static nerf::DeviceMask configured{0};
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_forward_bf16_kernel), mlp::B16_LDS_BYTES,
                                          configured, "nerf_mlp_forward_bf16: LDS attribute"))
        return rc;
    const int cus = nerf::device_cus();
    const int64_t ntiles = (M + TILE - 1) / TILE;
    hipLaunchKernelGGL(mlp_forward_bf16_kernel, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(256),
                       mlp::B16_LDS_BYTES, nerf::as_stream(stream), static_cast<const char *>(packed_bf16), pos,
                       view_dir, M, sigma, rgb);
    return nerf::check_launch("nerf_mlp_forward_bf16");
}
