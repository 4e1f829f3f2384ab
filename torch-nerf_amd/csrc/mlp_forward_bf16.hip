// bf16 inference variant of the fused encode + MLP kernel (BASELINE configs[2]: "bf16 MLP weights on
// MFMA").  Same algebra as mlp_forward.hip -- Y^T = W X^T, the D fragment of one layer is the B
// fragment of the next -- on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate):
//   * weights AND layer inputs are bf16 (round-to-nearest-even), accumulation, bias, the density row of fc_8,
//     fc_out and the sigmoid stay fp32; ReLU is applied to the packed bf16 pairs (except h7, which feeds the
//     density row unrounded)
//   * workgroup = 8 wavefronts = TWO per SIMD (256 registers each, no scratch), 32 samples per wavefront, 256 per
//     workgroup pass: a weight byte brought into LDS serves twice the samples of the 4-wave round-1 kernel.
//     That stream is what the kernel pays for: scripts/ldsdma_stream.hip (profiles/r02_ldsdma_stream.txt) runs the
//     bare MFMA + ds_read_b128 loop at ~1.9 PFLOP/s and the same loop with the 64 KiB / 256 samples weight stream
//     at ~1.5, with the shader clock, not the issue slots, giving way (the chip is power-bound under bf16 MFMA
//     load: 1.7 - 2.0 GHz instead of 2.4; the same kernel on all-zero weights runs 27 % faster)
//   * the two halves of the workgroup (waves 0-3 / 4-7, one of each per SIMD) run the same program ONE SUB-STEP
//     APART and every layer seam is split in two halves around the rendezvous, so that one wave's vector work
//     (bf16 packing, ReLU, bias, encodings) runs under its SIMD partner's MFMAs: the weight stream moves in 32-KiB
//     sub-steps (a quarter of a 256x256 layer) through a 4-slot LDS ring -- slot k%4 is read by the leading half
//     during global step k and by the trailing half during step k+1, and is refilled at step k+2; every wave
//     copies 4 KiB of every sub-step, its four DMA instructions spread between the MFMAs; one counted vmcnt wait
//     + one s_barrier per sub-step; 37 sub-steps per tile (mlp_layout.h), no filler
//   * every A fragment is one ds_read_b128 = 8 bf16 out of a conflict-free swizzled image, hand-issued three
//     fragments ahead
//   * positional encodings use one accurate sincos per channel and the double-angle recurrence
//     for the higher octaves (error <= 2^9 x 1e-7 = 5e-5, far below bf16 resolution 4e-3); they are
//     recomputed where needed (fc_in, fc_5, fc_9) instead of being held in registers, and packed octave by
//     octave so that only a few features are live at a time
// Parity target: PSNR against the reference's outputs (tests/test_gpu_bf16.py), not the 1e-5 bound.
#include <type_traits>

#include "mlp_device.h"
#include "net.h"

namespace {

using namespace mlp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WAVES = 8;
constexpr int TILE = 32 * WAVES;                 // samples per workgroup pass
constexpr int SUB_BYTES = B16_SUB_BYTES;         // 32 KiB: W[:, 64 k-values] of a 256-row layer
constexpr int RING = 4;
constexpr int SUBS_PER_TILE = B16_SUBS;          // 1 (fc_in) + 16 (fc_1..4) + 1 + 4 (fc_5) + 8 (fc_6,7) + 4 (fc_8) + 3 (fc_9)
constexpr int PIECES = SUB_BYTES / 1024 / WAVES; // 1-KiB DMA pieces per wave per sub-step
constexpr int W8_LDS_BYTES = RING * SUB_BYTES + CONST_BYTES;
static_assert(PIECES == 4, "ring geometry");

struct SubPipe {
    const char *src_wave;  // stream base + wave * 4 KiB (wave-uniform)
    unsigned lane_off;     // lane * 16
    unsigned lds_wave;     // LDS address of ring slot 0 + wave * 4 KiB
    unsigned issued;       // sub-steps requested so far (ring slot = issued % RING)
    int issue_q;           // position in the tile, [0, SUBS_PER_TILE), of the next sub-step to request
    unsigned consumed;     // sub-steps this wave has consumed

    __device__ __forceinline__ void issue_piece(int p) const {
        lds_dma_16s(src_wave + issue_q * SUB_BYTES + p * 1024, lane_off,
                    lds_wave + (issued & (RING - 1)) * SUB_BYTES + p * 1024);
    }
    __device__ __forceinline__ void issue_done() {
        ++issued;
        issue_q = (issue_q + 1 == SUBS_PER_TILE) ? 0 : issue_q + 1;
    }
    // barrier k of the kernel: sub-step k has landed (this wave's pieces of k and k+1 are the only DMA it has
    // outstanding: all but the youngest PIECES must be back), and the slot of sub-step k-2 is free for k+2
    __device__ __forceinline__ void rendezvous() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(PIECES) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ unsigned acquire() {
        rendezvous();
        const unsigned off = (consumed & (RING - 1)) * SUB_BYTES;
        ++consumed;
        return off;
    }
    // a step on which this wave consumes nothing (the trailing half before its first tile, the leading half
    // after its last): it still copies its share of the next sub-step
    __device__ __forceinline__ void idle_step() {
        rendezvous();
#pragma unroll
        for (int p = 0; p < PIECES; ++p) issue_piece(p);
        issue_done();
    }
};

__device__ __forceinline__ bf16x8 lds_read_fragment16(unsigned lds_addr, int imm_offset) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(imm_offset));
    return v;
}

// acc[fb] += W[32 fb.., 32 k of this chunk] . B ; b[s] are the two k-steps (16 features each) of the 32-feature
// input block.  N_PIECES > 0: also issue DMA pieces 0 .. N_PIECES-1 of the next sub-step, evenly spread over the
// (s, fb) groups of this chunk.
template <int NFB, int N_PIECES>
__device__ __forceinline__ void mma_chunk16(f32x16 (&acc)[8], const bf16x8 (&b)[2], const char *chunk,
                                            const int (&offs)[2], const SubPipe &pipe) {
    constexpr int GROUPS = 2 * NFB, EVERY = N_PIECES > 0 ? GROUPS / N_PIECES : 1;
    static_assert(N_PIECES == 0 || GROUPS % N_PIECES == 0, "pieces must divide the groups");
    // A fragments three groups ahead in four rotating buffers (a bf16 MFMA lasts only 32 cycles, an LDS
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
        acc[fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(abuf[g & 3], b[s], acc[fb], 0, 0, 0);
        if (N_PIECES > 0 && g % EVERY == 0) pipe.issue_piece(g / EVERY);
    }
}

// acc[4q..4q+3] <- bias_blk[8 q + 4 h ..]: the C fragment of one feature block starts as the bias
__device__ __forceinline__ void load_bias_block(f32x16 &acc, const float *bias_blk, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(bias_blk + 8 * q + 4 * h);
        acc[4 * q + 0] = v.x;
        acc[4 * q + 1] = v.y;
        acc[4 * q + 2] = v.z;
        acc[4 * q + 3] = v.w;
    }
}

// ReLU of 8 packed bf16 values: as 16-bit integers, negative floats are negative and non-negative floats are
// non-negative, so max(., 0) is the ReLU (-0 -> +0, negative NaN payloads -> 0, like v_max_f32 in IEEE mode)
__device__ __forceinline__ void relu_packed(bf16x8 &v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) asm("v_pk_max_i16 %0, %1, 0" : "=v"(w[j]) : "v"(w[j]));
    v = __builtin_bit_cast(bf16x8, w);
}

// 16 fp32 values of one D-fragment block -> the two bf16 B fragments (k-steps) of the next layer
__device__ __forceinline__ void pack_block(const f32x16 &x, bf16x8 (&frag)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) frag[s][e] = (__bf16)x[8 * s + e];
}

// B fragments of the first 32 NBLK encoding features of one sample for lane half h, NBLK blocks x 2 k-steps.
// Feature k of PositionalEncoder(3, LEVELS, include_input=True) (positional_encoder.py:83-88) is x,y,z for k < 3 and
// sin / cos(2^f v_c) at k = 3 + 6 f + 3 t + c; one accurate sincos per channel, the double-angle recurrence for the
// higher octaves.  Word j of fragment (blk, s) packs features k0, k0 + 1 with
// k0 = 32 blk + 8 (2 s + (j >> 1)) + 2 (j & 1) + 4 h.  Octave by octave, each word is emitted as soon as its (at
// most four) candidate features exist, so that only ~2 octaves of features are live at any time (the first
// version evaluated all 63 features first: 64 live floats on top of 128 accumulators spill at 256 registers).
// RT: `levels` <= LEVELS octaves at run time (the features of the others are zero, like the weight columns behind
// them in the stream); INC = include_input.  The shipped configuration is <10 | 4, ., true, false>.
template <int LEVELS, int NBLK, bool INC = true, bool RT = false>
__device__ __forceinline__ void encode_frags(float x, float y, float z, int h, bf16x8 (&frag)[NBLK][2], int levels = LEVELS) {
    constexpr int NF = 32 * NBLK, RAW = INC ? 3 : 0, KMAX = RAW + 6 * LEVELS;
    float F[NF + 8];   // static indices only: the compiler keeps just the live window in registers
#pragma unroll
    for (int k = 0; k < NF + 8; ++k) F[k] = 0.0f;
    const float v[3] = {x, y, z};
    float sn[3], cs[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (INC) F[c] = v[c];
        sincos_cw<false>(v[c], sn[c], cs[c]);   // |coordinate| itself is the only argument: no large-range issue below 3e4
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 words[NBLK][2];
    int emitted = 0;   // features [0, emitted) have been packed
    auto emit_upto = [&](int ready) {   // every word whose features are all < ready (ready: features [0, ready) final)
#pragma unroll
        for (int k0 = 0; k0 < NF; k0 += 8) {   // a word group = features k0 .. k0+7: words (k0, k0+1 | k0+4, k0+5), (k0+2, k0+3 | k0+6, k0+7)
            if (k0 >= emitted && k0 + 8 <= ready) {
                const int blk = k0 >> 5, q8 = (k0 & 31) >> 3;   // q8 = 2 s + (j >> 1)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float lo = h ? F[k0 + 4 + 2 * jj] : F[k0 + 2 * jj];
                    const float hi = h ? F[k0 + 5 + 2 * jj] : F[k0 + 1 + 2 * jj];
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    const bf16x2 pk = {(__bf16)lo, (__bf16)hi};
                    words[blk][q8 >> 1][2 * (q8 & 1) + jj] = __builtin_bit_cast(unsigned, pk);
                }
            }
        }
        emitted = ready & ~7;
    };
#pragma unroll
    for (int f = 0; f < LEVELS; ++f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool on = !RT || f < levels;      // (wave-uniform)
            if (RAW + 6 * f + c < NF) F[RAW + 6 * f + c] = on ? sn[c] : 0.0f;
            if (RAW + 6 * f + 3 + c < NF) F[RAW + 6 * f + 3 + c] = on ? cs[c] : 0.0f;
            const float s2 = 2.0f * sn[c] * cs[c];
            cs[c] = fmaf(-2.0f * sn[c], sn[c], 1.0f);
            sn[c] = s2;
        }
        emit_upto(RAW + 6 * (f + 1) < NF ? RAW + 6 * (f + 1) : NF);
    }
    emit_upto(NF);   // zero tail beyond KMAX
    static_assert(KMAX <= NF, "encoding wider than the fragment blocks");
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
        for (int s = 0; s < 2; ++s) frag[blk][s] = __builtin_bit_cast(bf16x8, words[blk][s]);
}

// MODE 0: the shipped encoders (compile-time recurrences; BASELINE configs[2]); 1 / 2: PositionalEncoders of run-time
// levels (<= 10 for the position, <= 4 | 5 for the direction) with / without include_input -- every other network of
// the fused family (runner_utils.py:584-612)
template <int MODE>
__global__ __launch_bounds__(64 * WAVES, 1) void mlp_forward_bf16_kernel(const Net net, const char *__restrict__ packed,
                                                                          const float *__restrict__ pos,
                                                                          const float *__restrict__ dir, int64_t M,
                                                                          float *__restrict__ sigma_out,
                                                                          float *__restrict__ rgb_out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool trailing = wave >= WAVES / 2;   // this half runs one sub-step behind the other
    const int i = lane & 31, h = lane >> 5;
    // const block FIRST (LDS offsets < 13 KiB fit the 16-bit offset field of ds_read: one address register for
    // all bias / head reads instead of one per block), ring behind it
    float *cb_ = reinterpret_cast<float *>(lds);
    const char *ring = lds + CONST_BYTES;
    for (int e = tid; e < CONST_FLOATS / 4; e += 64 * WAVES)
        reinterpret_cast<f32x4 *>(cb_)[e] = reinterpret_cast<const f32x4 *>(packed)[e];

    int offs[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) offs[s] = b16_frag_offset(i, 2 * s + h);

    SubPipe pipe;
    pipe.src_wave = packed + CONST_BYTES + wave * (PIECES * 1024);
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)CONST_BYTES +
                    (unsigned)wave * (PIECES * 1024u);
    pipe.issued = 0;
    pipe.issue_q = 0;
    pipe.consumed = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // sub-steps 0 and 1 are in flight before the first rendezvous
#pragma unroll
        for (int p = 0; p < PIECES; ++p) pipe.issue_piece(p);
        pipe.issue_done();
    }
#ifdef X_B16_NOLAG
    const bool lag = false;
#else
    const bool lag = true;
#endif
    if (trailing && lag) pipe.idle_step();
#ifdef X_B16_SETPRIO
    if (trailing) __builtin_amdgcn_s_setprio(1);
#endif

    const int64_t ntiles = (M + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t m = tile * TILE + wave * 32 + i;
        const bool valid = m < M;
        const int64_t mc = valid ? m : M - 1;
        float raw[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            raw[c] = pos[3 * mc + c];
            raw[3 + c] = dir[3 * mc + c];
        }

        f32x16 acc[8];      // [output feature block]
        bf16x8 act[8][2];   // [32-feature input block][k-step]
        float sigma_pre = 0.0f;

        // encoded position as B fragments, evaluated on demand (fc_in and the fc_5 skip connection)
        auto position_frags = [&](bf16x8 (&pe)[2][2]) {  // [blk][s]
            if (MODE == 0) encode_frags<DEFAULT_NET.l_pos, 2>(raw[0], raw[1], raw[2], h, pe);
            else encode_frags<10, 2, MODE == 1, true>(raw[0], raw[1], raw[2], h, pe, net.l_pos);
        };
        // one sub-step = two 32-feature input blocks against all NFB output blocks
        auto sub_step = [&](const char *w, const bf16x8 (&b0)[2], const bf16x8 (&b1)[2], auto nfb_tag) {
            constexpr int NFB = decltype(nfb_tag)::value;
            mma_chunk16<NFB, PIECES>(acc, b0, w, offs, pipe);
            mma_chunk16<NFB, 0>(acc, b1, w + B16_CHUNK_BYTES, offs, pipe);
            pipe.issue_done();
        };
        // Layer seam, one HALF (feature blocks 4 HALF_IX .. 4 HALF_IX + 3) per call: activation of the finished
        // layer -> packed bf16 inputs of the next one, and the accumulator block restarts from the next layer's
        // bias (the C fragment starts as the bias).  The first half runs BEFORE the rendezvous of the next
        // layer's first sub-step, the second half behind it, so that neither side of the barrier carries more
        // than half a seam: the SIMD partner (one sub-step ahead or behind) covers both with its MFMAs.
        // ReLU is applied to the PACKED bf16 pairs (v_pk_max_i16 against 0: a negative bf16 is a negative int16),
        // 8 instead of 16 instructions per block; the density row of fc_8 needs the unrounded fp32 h7, so that one
        // seam keeps the fp32 ReLU.  Flags are compile-time: a run-time flag puts a branch into every block and hipcc
        // then spills around it -- and a spill reload is a VMEM load whose vmcnt wait also waits for the weight DMA.
        auto seam_half = [&](auto half_tag, auto relu_tag, auto density_tag, auto next_blocks_tag, const float *next_bias) {
            constexpr int HALF_IX = decltype(half_tag)::value, NEXT_BLOCKS = decltype(next_blocks_tag)::value;
            constexpr bool RELU = decltype(relu_tag)::value, DENSITY = decltype(density_tag)::value;
#pragma unroll
            for (int fb = 4 * HALF_IX; fb < 4 * HALF_IX + 4; ++fb) {
                f32x16 x;
#ifdef X_B16_F32RELU
                constexpr bool PACKED_RELU = false;
#else
                constexpr bool PACKED_RELU = RELU && !DENSITY;
#endif
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = (RELU && !PACKED_RELU) ? relu1(acc[fb][r]) : acc[fb][r];
                if (DENSITY) sigma_pre += block_dot(cb_ + CB_W8ROW0 + 32 * fb, x, h);
                pack_block(x, act[fb]);
                if (PACKED_RELU) {
                    relu_packed(act[fb][0]);
                    relu_packed(act[fb][1]);
                }
                // block by block (one fence each): hipcc otherwise hoists all bias reads of the layer to the
                // top of the seam and spills the accumulators they are going to replace
                if (fb < NEXT_BLOCKS) load_bias_block(acc[fb], next_bias + 32 * fb, h);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        typedef std::integral_constant<int, 8> Full;
        typedef std::integral_constant<int, 4> Half;
        typedef std::integral_constant<int, 0> First;
        typedef std::integral_constant<int, 1> Second;
        typedef std::true_type Yes;
        typedef std::false_type No;
        // sub-steps 2..4 of a 256-wide layer whose first sub-step (act[0], act[1]) has been issued
        auto rest_of_layer = [&]() {
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const char *w = ring + pipe.acquire();
                sub_step(w, act[2 * q], act[2 * q + 1], Full());
            }
        };
        // a plain 256 -> 256 layer l (ReLU of layer l-1 in its seam)
        auto plain_layer = [&](int l) {
#ifndef X_B16_NOSPLIT
            seam_half(First(), Yes(), No(), Full(), cb_ + CB_BIAS + l * 256);
            const char *w = ring + pipe.acquire();
#else
            const char *w = ring + pipe.acquire();
            seam_half(First(), Yes(), No(), Full(), cb_ + CB_BIAS + l * 256);
#endif
            seam_half(Second(), Yes(), No(), Full(), cb_ + CB_BIAS + l * 256);
            sub_step(w, act[0], act[1], Full());
            rest_of_layer();
        };

        // ---- fc_in (nerf.py:102): sub-step 0
        {
            bf16x8 pe[2][2];
            position_frags(pe);
            load_bias<8>(acc, cb_ + CB_BIAS, h);
            const char *w = ring + pipe.acquire();
            sub_step(w, pe[0], pe[1], Full());
        }
        // ---- fc_1 .. fc_4 (:103-106)
        for (int l = 1; l <= 4; ++l) plain_layer(l);
        // ---- fc_5 on cat([pos, x]) (:108): position FIRST
        {
            // encode first while layer 4's outputs still sit in the accumulators (its inputs are dead), then
            // finish layer 4 into `act`
            bf16x8 pe[2][2];
            position_frags(pe);
            seam_half(First(), Yes(), No(), Full(), cb_ + CB_BIAS + 5 * 256);
            const char *w = ring + pipe.acquire();
            seam_half(Second(), Yes(), No(), Full(), cb_ + CB_BIAS + 5 * 256);
            sub_step(w, pe[0], pe[1], Full());
            w = ring + pipe.acquire();
            sub_step(w, act[0], act[1], Full());
            rest_of_layer();
        }
        // ---- fc_6, fc_7 (:109-110)
        for (int l = 6; l <= 7; ++l) plain_layer(l);
        // ---- fc_8 (:113): rows 1..256 on the matrix pipe; h7 (unrounded fp32) feeds the density row on the vector ALU
        {
            seam_half(First(), Yes(), Yes(), Full(), cb_ + CB_BIAS8);
            const char *w = ring + pipe.acquire();
            seam_half(Second(), Yes(), Yes(), Full(), cb_ + CB_BIAS8);
            sub_step(w, act[0], act[1], Full());
            rest_of_layer();
        }
        // ---- fc_9 on cat([x[:,1:], view_dir]) (:116-118), 128 rows: three 8-KiB chunks per sub-step; fc_8 has no
        // ReLU (:113)
        {
            seam_half(First(), No(), No(), Half(), cb_ + CB_BIAS9);
            const char *w = ring + pipe.acquire();
            seam_half(Second(), No(), No(), Half(), cb_ + CB_BIAS9);
            mma_chunk16<4, PIECES>(acc, act[0], w, offs, pipe);
            mma_chunk16<4, 0>(acc, act[1], w + B16_HALF_CHUNK_BYTES, offs, pipe);
            mma_chunk16<4, 0>(acc, act[2], w + 2 * B16_HALF_CHUNK_BYTES, offs, pipe);
            pipe.issue_done();
            w = ring + pipe.acquire();
            mma_chunk16<4, PIECES>(acc, act[3], w, offs, pipe);
            mma_chunk16<4, 0>(acc, act[4], w + B16_HALF_CHUNK_BYTES, offs, pipe);
            mma_chunk16<4, 0>(acc, act[5], w + 2 * B16_HALF_CHUNK_BYTES, offs, pipe);
            pipe.issue_done();
            bf16x8 de[1][2];
            if (MODE == 0) encode_frags<DEFAULT_NET.l_dir, 1>(raw[3], raw[4], raw[5], h, de);
            else encode_frags<(MODE == 1 ? 4 : 5), 1, MODE == 1, true>(raw[3], raw[4], raw[5], h, de, net.l_dir);
            w = ring + pipe.acquire();
            mma_chunk16<4, PIECES>(acc, act[6], w, offs, pipe);
            mma_chunk16<4, 0>(acc, act[7], w + B16_HALF_CHUNK_BYTES, offs, pipe);
            mma_chunk16<4, 0>(acc, de[0], w + 2 * B16_HALF_CHUNK_BYTES, offs, pipe);
            pipe.issue_done();
        }

        // ---- ReLU(fc_9), fc_out, sigmoid (:118-119) and sigma = relu(x[:,0]) (:115), fp32 vector ALU
        {
            float y[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                f32x16 x;
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = relu1(acc[fb][r]);
#pragma unroll
                for (int c = 0; c < 3; ++c) y[c] += block_dot(cb_ + CB_WOUT + c * HALF + 32 * fb, x, h);
            }
            const float sp = sigma_pre + __shfl_xor(sigma_pre, 32, WAVE);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float p = y[c] + __shfl_xor(y[c], 32, WAVE);
                y[c] = 1.0f / (1.0f + expf(-(p + cb_[CB_SCALARS + 1 + c])));
            }
            if (valid && h == 0) {
                sigma_out[m] = fmaxf(sp + cb_[CB_SCALARS], 0.0f);
                rgb_out[3 * m + 0] = y[0];
                rgb_out[3 * m + 1] = y[1];
                rgb_out[3 * m + 2] = y[2];
            }
        }
    }
    if (!trailing && lag) pipe.idle_step();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

NERF_API int nerf_mlp_forward_bf16(const nerf_net_t *net_abi, const void *packed_bf16, const float *pos,
                                   const float *view_dir, int64_t M, float *sigma, float *rgb, nerf_stream_t stream) {
    mlp::Net net;
    if (int rc = nerf::fused_net(net_abi, net, "nerf_mlp_forward_bf16")) return rc;
    if (!nerf::raw_inputs_ok(net) || net.inc_pos != net.inc_dir)
        return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_mlp_forward_bf16: the kernel encodes raw points: nerf_net_t needs the "
                                                "levels of both PositionalEncoders (one include_input for both)");
    NERF_REQUIRE(M >= 0, "nerf_mlp_forward_bf16: negative M");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(packed_bf16 && pos && view_dir && sigma && rgb, "nerf_mlp_forward_bf16: null pointer");
    const int mode = net.is_default() ? 0 : net.inc_pos ? 1 : 2;
    auto kern = mode == 0 ? mlp_forward_bf16_kernel<0> : mode == 1 ? mlp_forward_bf16_kernel<1> : mlp_forward_bf16_kernel<2>;
    static nerf::DeviceMask configured[3] = {{0}, {0}, {0}};
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), W8_LDS_BYTES, configured[mode],
                                          "nerf_mlp_forward_bf16: LDS attribute"))
        return rc;
    const int cus = nerf::device_cus();
    const int64_t ntiles = (M + TILE - 1) / TILE;
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(64 * WAVES), W8_LDS_BYTES,
                       nerf::as_stream(stream), net, static_cast<const char *>(packed_bf16), pos, view_dir, M, sigma, rgb);
    return nerf::check_launch("nerf_mlp_forward_bf16");
}
