// fp32-grade inference on the f16 matrix pipe: the fused encode + MLP kernel with every operand SPLIT in two f16 parts.
//
// Why: the fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the vector rate -- 157 TFLOP/s, 1/16 of the f16 / bf16 matrix
// pipe -- and the fp32 kernels sit at 0.94 of that peak.  An fp32 value is hi + lo with hi = f16(x), lo = f16(x - hi)
// to 22 significand bits (subnormal low parts are kept by the pipe: scripts/f16_split_probe.hip), the part products
// are exact in the pipe's fp32 accumulation, and lo.hi + hi.lo + hi.hi reproduces an fp32 dot product to the fp32
// kernel's own error class (K = 256: 1.0 - 2.5e-7 of sum |a b| against 1.1 - 1.3e-7 for the fp32 MFMA chain,
// profiles/r06_f16_split_probe.txt).  Three products at 16x the rate: the bound of north_star (1e-5 abs on sigma, rgb,
// pixels) is met with the margin of the fp32 kernels (scripts/split_emulate.py on goldens F5 / F7 / F11: <= 9e-7), which
// bf16 parts cannot do in three products (1e-5: two bf16 parts carry 16 bits).
//
// Same algebra as mlp_forward_bf16.hip -- Y^T = W X^T, the D fragment of one layer is the B fragment of the next -- with:
//   * v_mfma_f32_16x16x32_f16, SIXTEEN samples per wavefront: a layer's state per lane is 64 accumulators + 32 + 32
//     registers of packed hi / lo activations, so that two wavefronts per SIMD fit (256 registers each) and one wave's
//     seam (scale, ReLU, split, bias) runs under its SIMD partner's MFMAs.  (32 samples per wave would need 128 + 64 +
//     64 = 256 registers before the first temporary.)
//   * workgroup = 8 wavefronts = 128 samples per pass; weights scaled per layer by a power of two, split at pack time
//     (mlp_pack.hip), streamed in 32-KiB sub-steps [hi image | lo image] of one 32-wide k-block through the bf16 kernel's
//     4-slot LDS ring; 73 sub-steps per tile.  An A fragment is one ds_read_b128 out of an image swizzled for the lane
//     sets ds_read_b128 serves together (mlp_layout.h:f2_frag_offset; zero bank conflicts, profiles/r06_pmc_f16x2.txt);
//     the hi fragment feeds two MFMAs, the lo fragment one: 2 KiB of LDS reads per 3 MFMAs
//   * encodings in fp32 (one accurate sincos per needed feature, the library path for huge arguments, like the fp32
//     kernels), split like every other activation; biases ride pre-scaled in the C fragment; the density row of fc_8
//     (from the unsplit fp32 h7), fc_out and the sigmoid stay fp32 on the vector ALU
// Every NeRF(pos_dim <= 128, view_dir_dim <= 64, 256) behind two PositionalEncoders (run-time levels / include_input): the
// fused family and the wider inputs of coord_encode_level 11..20 / dir_encode_level 5..10.  Inference only.
// Range: an input, encoding or activation beyond +-65504 cannot be split; the kernel tracks the largest magnitude it
// splits per sample and returns NaN for such a sample (never a silent wrong value).
#include <type_traits>

#include "mlp_device.h"
#include "net.h"

namespace {

using namespace mlp;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 8;
constexpr int WSAMPLES = 16;                     // samples per wavefront (the N of 16x16x32)
constexpr int TILE = WSAMPLES * WAVES;           // samples per workgroup pass
constexpr int SUB_BYTES = F2_SUB_BYTES;
constexpr int RING = 4;
constexpr int PIECES = SUB_BYTES / 1024 / WAVES; // 1-KiB DMA pieces per wave per sub-step
constexpr int F2_LDS_BYTES = RING * SUB_BYTES + CONST_BYTES;
static_assert(PIECES == 4, "ring geometry");

// (the bf16 kernel's pipe: see mlp_forward_bf16.hip for the slot / barrier protocol)
struct SubPipe {
    const char *src_wave;  // stream base + wave * 4 KiB (wave-uniform)
    unsigned lane_off;     // lane * 16
    unsigned lds_wave;     // LDS address of ring slot 0 + wave * 4 KiB
    unsigned issued;       // sub-steps requested so far (ring slot = issued % RING)
    int issue_q;           // position in the tile, [0, subs_per_tile), of the next sub-step to request
    unsigned consumed;     // sub-steps this wave has consumed
    int subs_per_tile;     // F2Layout::subs() of the network
    int stores;            // (record mode) VMEM stores this wave has issued since its last rendezvous

    __device__ __forceinline__ void issue_piece(int p) const {
        lds_dma_16s(src_wave + issue_q * SUB_BYTES + p * 1024, lane_off,
                    lds_wave + (issued & (RING - 1)) * SUB_BYTES + p * 1024);
    }
    __device__ __forceinline__ void issue_done() {
        ++issued;
        issue_q = (issue_q + 1 == subs_per_tile) ? 0 : issue_q + 1;
    }
    __device__ __forceinline__ void rendezvous() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(PIECES) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // Record mode: the wave also has plane stores in flight.  gfx9 retires loads and stores through ONE in-order vmcnt, so
    // "everything but the operations issued since the last rendezvous" is exact: those are this sub-step's PIECES DMA
    // pieces + `stores` stores, all younger than the pieces of the sub-step being acquired (issued one rendezvous
    // earlier).  s_waitcnt takes an immediate: a scalar jump table over the store count (capped low = waits for more).
    __device__ __forceinline__ void rendezvous_recording() {
        const int n = stores;
        stores = 0;
        if (n <= 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(PIECES) : "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(PIECES + 1) : "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(PIECES + 2) : "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(PIECES + 3) : "memory");
        else if (n <= 5) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(PIECES + 4) : "memory");
        else if (n <= 8) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(PIECES + 6) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(PIECES + 9) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    template <bool RECORDING>
    __device__ __forceinline__ unsigned acquire_as() {
        if (RECORDING) rendezvous_recording(); else rendezvous();
        const unsigned off = (consumed & (RING - 1)) * SUB_BYTES;
        ++consumed;
        return off;
    }
    __device__ __forceinline__ unsigned acquire() {
        rendezvous();
        const unsigned off = (consumed & (RING - 1)) * SUB_BYTES;
        ++consumed;
        return off;
    }
    __device__ __forceinline__ void idle_step() {
        rendezvous();
#pragma unroll
        for (int p = 0; p < PIECES; ++p) issue_piece(p);
        issue_done();
    }
};

__device__ __forceinline__ f16x8 lds_read_fragment16(unsigned lds_addr, int imm_offset) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(imm_offset));
    return v;
}

// acc[fb] += W'[16 fb .. 16 fb + 15][this 32-wide k-block] . x  for fb < NFB, W' = hi + lo and x = bhi + blo:
//   acc += lo.bhi ; acc += hi.blo ; acc += hi.bhi      (small terms first)
// `addr` = LDS address of the k-block's hi image + this lane's fragment offset; the lo image sits LO_OFF bytes behind.
// Output blocks go two at a time (six MFMAs, the dependent ones three apart), their four A fragments read two pairs ahead
// in three rotating buffer sets; N_PIECES > 0: DMA pieces 0 .. N_PIECES-1 of the next sub-step ride between the pairs.
template <int NFB, int N_PIECES, int HI_OFF, int LO_OFF>
__device__ __forceinline__ void mma_kblock(f32x4 (&acc)[16], const f16x8 &bhi, const f16x8 &blo, unsigned addr,
                                           const SubPipe &pipe) {
#ifndef X_F2_AHEAD
#define X_F2_AHEAD 2          // A/B knob: pairs of output blocks whose A fragments are in flight ahead of the MFMAs
#endif
    constexpr int PAIRS = NFB / 2, AHEAD = X_F2_AHEAD, NBUF = AHEAD + 1, EVERY = N_PIECES > 0 ? PAIRS / N_PIECES : 1;
    static_assert(NFB % 2 == 0 && (N_PIECES == 0 || PAIRS % N_PIECES == 0), "pairs of output blocks; pieces divide them");
    f16x8 ah[NBUF][2], al[NBUF][2];
    auto fetch = [&](int p) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ah[p % NBUF][j] = lds_read_fragment16(addr, HI_OFF + (2 * p + j) * 1024);
            al[p % NBUF][j] = lds_read_fragment16(addr, LO_OFF + (2 * p + j) * 1024);
        }
    };
#pragma unroll
    for (int p = 0; p < AHEAD && p < PAIRS; ++p) fetch(p);
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
        // pairs p+1 .. p+AHEAD-1 may still be in flight: four reads each
        __builtin_amdgcn_sched_barrier(0);
        {
            constexpr int LEFT[3] = {0, 4, 8};
            const int younger = (PAIRS - 1 - p) < (AHEAD - 1) ? (PAIRS - 1 - p) : (AHEAD - 1);
            if (LEFT[younger] == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else if (LEFT[younger] == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if (p + AHEAD < PAIRS) fetch(p + AHEAD);
        const int b = p % NBUF, f0 = 2 * p, f1 = 2 * p + 1;
        acc[f0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b][0], bhi, acc[f0], 0, 0, 0);
        acc[f1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b][1], bhi, acc[f1], 0, 0, 0);
        acc[f0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b][0], blo, acc[f0], 0, 0, 0);
        acc[f1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b][1], blo, acc[f1], 0, 0, 0);
        if (N_PIECES > 0 && p % EVERY == 0) pipe.issue_piece(p / EVERY);
        acc[f0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b][0], bhi, acc[f0], 0, 0, 0);
        acc[f1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b][1], bhi, acc[f1], 0, 0, 0);
    }
}

// Record mode (training forward): where this lane's four features of a 16-feature output block go in a plane of the fused
// family's activation record (mlp_layout.h: TF layout; mlp_backward.hip reads it).  Lane (n, g) of wave w holds features
// 16 fb + 4 g + e (e = 0..3) of sample m = 128 tile + 16 w + n: 32-feature block fb >> 1, quarter q = 2 (fb & 1) + (g >> 1),
// lane half h = g & 1 -- one 16-byte store per block, 2 x 512 contiguous bytes per wavefront store.
struct Recorder {
    char *base;                 // the record
    int64_t MP;                 // padded rows
    int64_t tile32;             // m >> 5 (wave-uniform)
    unsigned off_even, off_odd; // byte offset inside the (tile, 32-feature block) KiB run for even / odd fb
    unsigned mask_unit;         // (2 (m & 31) + h) * 16: this lane's mask word inside the tile's 1-KiB mask run
    int g;
    __device__ __forceinline__ void open(float *saved, int64_t MP_, int64_t m, int g_) {
        base = reinterpret_cast<char *>(saved);
        MP = MP_;
        tile32 = m >> 5;
        g = g_;
        const unsigned i = (unsigned)(m & 31), h = (unsigned)(g_ & 1), q0 = (unsigned)(g_ >> 1);
        off_even = (q0 << 10) + 16u * ((2u * i + h) ^ (2u * q0));
        off_odd = ((q0 + 2u) << 10) + 16u * ((2u * i + h) ^ (2u * (q0 + 2u)));
        mask_unit = (2u * i + h) * 16u;
    }
    // plane at float offset `plane` (x MP already applied), `width` floats per sample; block fb of 16 features
    __device__ __forceinline__ void store(int64_t plane, int width, int fb, const f32x4 &x, SubPipe &pipe) const {
        char *p = base + 4 * (plane + tile32 * 32 * width) + (fb >> 1) * 4096 + ((fb & 1) ? off_odd : off_even);
        // Nontemporal: these planes are read next by ANOTHER kernel, 8 GB of stores later -- kept out of the L2's way they cost
        // 2 % less (record forward 2.69 -> 2.63 ms, reverse chain alike; sc1 loses 5 %).  Without any store the record
        // forward takes 2.12 ms, and with the same stores aimed at 1 MiB that stays in the L2 (no HBM traffic) 2.54: four
        // fifths of their price is on the chip -- 1 KiB per instruction through the CU's 64 B / clock store path, in the
        // layer seams where the matrix pipe idles anyway, and their share of the in-order vmcnt waits -- not the HBM
        // (scripts/hbm_write_probe.hip: 6.3 TB/s for this pattern).  (s_nop 1: two wait states before anything may overwrite
        // the data registers.)
        asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
        ++pipe.stores;
    }
    // dword fb >> 2 of this lane's half of the (sample, h) mask word: bit 16 ((fb >> 1) & 1) + e + 4 q  <-  x[e] > 0
    // (x is post-ReLU: > 0 <=> bit pattern != 0, mlp_device.h:save_mask)
    __device__ __forceinline__ void mask_bits(int fb, const f32x4 &x, unsigned (&words)[4]) const {
        unsigned w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned t;
            asm("v_min_u32 %0, 1, %1" : "=v"(t) : "v"(x[e]));
            w |= t << e;
        }
        words[fb >> 2] |= w << (16 * ((fb >> 1) & 1) + 8 * (fb & 1) + 4 * (g >> 1));
    }
    // mask plane l (0..7: h0..h7, 8: h9): lanes g and g ^ 2 hold the two halves of a word; lanes g < 2 store it
    __device__ __forceinline__ void store_mask(int l, const unsigned (&words)[4], SubPipe &pipe) const {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v;
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = words[d] | (unsigned)__shfl_xor((int)words[d], 32, WAVE);
        if (g < 2) {
            char *p = base + 4 * (pl_masks(MP) + (int64_t)l * MP * 8) + tile32 * 1024 + mask_unit;
            *reinterpret_cast<u32x4 *>(p) = v;
        }
        ++pipe.stores;
    }
};

// x (fp32, four features of one output block) -> elements 4 half .. 4 half + 3 of the hi / lo B fragments
// `amax` follows the largest |x| this lane has split: beyond 65504 the hi part is inf, inf - inf = NaN in the next
// accumulator and the ReLU behind it turns that into 0 -- a finite, wrong output.  The kernel poisons such a sample.
__device__ __forceinline__ void split4(const f32x4 &x, f16x8 &hi, f16x8 &lo, int half, float &amax) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(x[0]), "v"(x[1]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(x[2]), "v"(x[3]));
#ifndef X_F2_PLAINSPLIT    // the residual x - hi as ONE v_fma_mix_f32 per value, hi read as an f16 operand in place (hipcc's own
                          // sequence converts hi back to fp32 first: 5 instead of 3.5 instructions per value, 3 % of the kernel --
                          // profiles/r06_ab_f16x2.txt; X_F2_PLAINSPLIT builds that one for A/B)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 H = __builtin_bit_cast(u32x4, hi), L = __builtin_bit_cast(u32x4, lo);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        unsigned hp, lp;
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(x[2 * j]), "v"(x[2 * j + 1]));
        // d = 1.0 * x - hi:  v_fma_mix_f32 d, x, 1.0, -hi(f16, low | high half of hp)
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x[2 * j]), "v"(hp));
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x[2 * j + 1]), "v"(hp));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lp) : "v"(r0), "v"(r1));
        H[2 * half + j] = hp;
        L[2 * half + j] = lp;
    }
    hi = __builtin_bit_cast(f16x8, H);
    lo = __builtin_bit_cast(f16x8, L);
    return;
#endif
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const _Float16 h = (_Float16)x[r];
        hi[4 * half + r] = h;
        lo[4 * half + r] = (_Float16)(x[r] - (float)h);      // exact difference (compiled with -ffp-contract=off)
    }
}

// B fragments (hi, lo) of NKB 32-wide k-blocks of PositionalEncoder(3, levels, include_input).encode((x, y, z)) for lane
// group g: element e of block kb is feature 32 kb + 16 (e>>2) + 4 g + (e&3) (positional_encoder.py:83-88), zero from
// `width` = out_dim on
// REC: also store the fp32 encodings into plane `plane` of the record (by reference + flag: a Recorder whose address is
// taken conditionally ends up in scratch)
template <int NKB, bool EXACT, bool REC>
__device__ __forceinline__ void encode_split(float x, float y, float z, int g, int width, int include_input,
                                             f16x8 (&hi)[NKB], f16x8 (&lo)[NKB], float &amax,
                                             const Recorder &rec, int64_t plane, SubPipe &pipe) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = enc_feature<EXACT>(32 * kb + 16 * half + 4 * g + r, x, y, z, width, include_input);
            if (REC) rec.store(plane, 32 * NKB, 2 * kb + half, v, pipe);       // PL_PE / PL_DE: the fp32 encodings
            split4(v, hi[kb], lo[kb], half, amax);
        }
}

// NPOS (2 | 3 | 4) position k-blocks, NDIR (1 | 2) direction k-blocks: <2, 1> is the fused family (the shipped 63 / 27 and
// every coord_encode_level <= 10 / dir_encode_level <= 4); the others serve the wider encoders the yaml can name
// SAVE (<2, 1> only): the TRAINING forward -- the same arithmetic, plus the fused family's activation record (post-ReLU
// h0..h7, fc_8's output, h9, the fp32 encodings, ReLU bit planes) that mlp_backward.hip's fp32 kernels read
template <int NPOS, int NDIR, bool SAVE = false>
__global__ __launch_bounds__(64 * WAVES, 1) void mlp_forward_f16x2_kernel(const Net net, const char *__restrict__ packed,
                                                                           const float *__restrict__ pos,
                                                                           const float *__restrict__ dir, int64_t M,
                                                                           float *__restrict__ sigma_out,
                                                                           float *__restrict__ rgb_out,
                                                                           float *__restrict__ saved) {
    static_assert(!SAVE || (NPOS == 2 && NDIR == 1), "the record is the fused family's");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool trailing = wave >= WAVES / 2;   // this half runs one sub-step behind the other
    const int n = lane & 15, g = lane >> 4;    // sample of the wave's 16, lane group (rows 4 g .. 4 g + 3 of every D block)
    float *cb_ = reinterpret_cast<float *>(lds);
    const unsigned ring = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)CONST_BYTES;
    for (int e = tid; e < CONST_FLOATS / 4; e += 64 * WAVES)
        reinterpret_cast<f32x4 *>(cb_)[e] = reinterpret_cast<const f32x4 *>(packed)[e];

    // A fragment of output block fb, lane (row n of the block, lane group g): row 16 fb + n, logical slot g
    const unsigned frag = ring + (unsigned)f2_frag_offset(n, g);

    SubPipe pipe;
    pipe.src_wave = packed + CONST_BYTES + wave * (PIECES * 1024);
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave = ring + (unsigned)wave * (PIECES * 1024u);
    pipe.issued = 0;
    pipe.issue_q = 0;
    pipe.consumed = 0;
    pipe.subs_per_tile = F2Layout{NPOS, NDIR}.subs();
    pipe.stores = 0;
    const int64_t MP = padded_rows(M);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // sub-steps 0 and 1 are in flight before the first rendezvous
#pragma unroll
        for (int p = 0; p < PIECES; ++p) pipe.issue_piece(p);
        pipe.issue_done();
    }
    // The bf16 kernel runs the two halves of the workgroup one sub-step apart so that one wave's seam falls under its SIMD
    // partner's MFMAs.  Here the waves drift apart on their own (a seam is 1/6 of a layer, not 1/3) and the in-phase
    // schedule measures 0.4 - 1.4 % faster (profiles/r06_ab_f16x2.txt); X_F2_LAG builds the lagged one for A/B.
#ifdef X_F2_LAG
    const bool lag = true;
#else
    const bool lag = false;
#endif
    if (trailing && lag) pipe.idle_step();

    const int64_t ntiles = (M + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t m = tile * TILE + wave * WSAMPLES + n;
        const bool valid = m < M;
        const int64_t mc = valid ? m : M - 1;
        float raw[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            raw[c] = pos[3 * mc + c];
            raw[3 + c] = dir[3 * mc + c];
        }
        // (wave-uniform: one sample with a huge argument sends the wave down the library path)
        const bool exact = __any(encoding_needs_exact(raw, net.l_pos, net.l_dir));

        f32x4 acc[16];                 // [16-feature output block]
        f16x8 act_hi[8], act_lo[8];    // [32-feature input block]
        f16x8 pe_hi[NPOS], pe_lo[NPOS];   // the encoded position: fc_in and the fc_5 skip connection
        f16x8 de_hi[NDIR], de_lo[NDIR];   // the encoded direction (fc_9): evaluated up front too, while nothing else is live
        float sigma_pre = 0.0f;
        float amax = 0.0f;             // largest magnitude split so far (f16 range check, see split4)
        Recorder rec = {};
        if (SAVE) rec.open(saved, MP, m, g);
        unsigned mwords[4] = {0u, 0u, 0u, 0u};
        // (the lane group as a value hipcc cannot reason about: the feature-index arithmetic of the encodings would
        // otherwise be hoisted out of the tile loop and held in ~50 registers across the whole MFMA stream)
        int ge = g;
        asm volatile("" : "+v"(ge));

        // one sub-step of a 256-row layer: one k-block against all 16 output blocks
        auto sub_step = [&](const f16x8 &bhi, const f16x8 &blo) {
            const unsigned a = frag + pipe.template acquire_as<SAVE>();
            mma_kblock<16, PIECES, 0, F2_IMAGE_BYTES>(acc, bhi, blo, a, pipe);
            pipe.issue_done();
        };
        // accumulators <- the (pre-scaled) bias of the next layer, blocks [FIRST, FIRST + COUNT)
        auto load_bias_blocks = [&](const float *bias, int first, int count) {
#pragma unroll
            for (int fb = 0; fb < 16; ++fb)
                if (fb >= first && fb < first + count) acc[fb] = *reinterpret_cast<const f32x4 *>(bias + 16 * fb + 4 * g);
        };
        // Layer seam, one HALF (output blocks 8 HALF_IX .. 8 HALF_IX + 7) per call: accumulators of the finished layer l ->
        // its activation (x 2^-s_l, ReLU) -> packed hi / lo inputs of the next layer; the accumulator block restarts from
        // the next layer's bias.  As in the bf16 kernel the first half runs BEFORE the rendezvous of the next layer's
        // first sub-step and the second half behind it.  DENSITY: h7 also feeds the density row of fc_8 in fp32.
        auto seam_half = [&](auto half_tag, auto relu_tag, auto density_tag, auto next_blocks_tag, int l, const float *next_bias) {
            constexpr int HALF_IX = decltype(half_tag)::value, NEXT_BLOCKS = decltype(next_blocks_tag)::value;
            constexpr bool RELU = decltype(relu_tag)::value, DENSITY = decltype(density_tag)::value;
            const float unscale = cb_[F2_CB_UNSCALE + l];
#pragma unroll
            for (int fb = 8 * HALF_IX; fb < 8 * HALF_IX + 8; ++fb) {
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[fb][r] * unscale;
                    x[r] = RELU ? relu1(v) : v;
                }
                if (SAVE) {      // layer l's activation (fc_8: its output, no ReLU) into the record; ReLU decisions as bits
                    rec.store(l < 8 ? pl_h(MP, l) : pl_y8(MP), 256, fb, x, pipe);
                    if (RELU) rec.mask_bits(fb, x, mwords);
                    if (RELU && fb == 15) {
                        rec.store_mask(l, mwords, pipe);
                        mwords[0] = mwords[1] = mwords[2] = mwords[3] = 0u;
                    }
                }
                if (DENSITY) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(cb_ + CB_W8ROW0 + 16 * fb + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sigma_pre = fmaf(w[r], x[r], sigma_pre);
                    asm volatile("" : "+v"(sigma_pre));   // here, not sunk to its use behind fc_9 (hipcc then parks all of h7 in scratch)
                }
                split4(x, act_hi[fb >> 1], act_lo[fb >> 1], fb & 1, amax);
                if (fb < NEXT_BLOCKS) acc[fb] = *reinterpret_cast<const f32x4 *>(next_bias + 16 * fb + 4 * g);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        typedef std::integral_constant<int, 16> Full;
        typedef std::integral_constant<int, 8> Half;
        typedef std::integral_constant<int, 0> First;
        typedef std::integral_constant<int, 1> Second;
        typedef std::true_type Yes;
        typedef std::false_type No;
        // a 256 -> 256 layer whose inputs are the outputs of layer `prev` (ReLU in the seam): 8 sub-steps
        auto plain_layer = [&](int prev, auto density_tag, const float *bias) {
            seam_half(First(), Yes(), density_tag, Full(), prev, bias);
            unsigned a = frag + pipe.template acquire_as<SAVE>();
            seam_half(Second(), Yes(), density_tag, Full(), prev, bias);
            mma_kblock<16, PIECES, 0, F2_IMAGE_BYTES>(acc, act_hi[0], act_lo[0], a, pipe);
            pipe.issue_done();
#pragma unroll
            for (int kb = 1; kb < 8; ++kb) sub_step(act_hi[kb], act_lo[kb]);
        };

        // ---- fc_in (nerf.py:102): sub-steps 0, 1
        if (exact) {
            encode_split<NPOS, true, SAVE>(raw[0], raw[1], raw[2], ge, net.e_pos, net.inc_pos, pe_hi, pe_lo, amax, rec, pl_pe(MP), pipe);
            encode_split<NDIR, true, SAVE>(raw[3], raw[4], raw[5], ge, net.e_dir, net.inc_dir, de_hi, de_lo, amax, rec, pl_de(MP), pipe);
        } else {
            encode_split<NPOS, false, SAVE>(raw[0], raw[1], raw[2], ge, net.e_pos, net.inc_pos, pe_hi, pe_lo, amax, rec, pl_pe(MP), pipe);
            encode_split<NDIR, false, SAVE>(raw[3], raw[4], raw[5], ge, net.e_dir, net.inc_dir, de_hi, de_lo, amax, rec, pl_de(MP), pipe);
        }
        load_bias_blocks(cb_ + CB_BIAS, 0, 16);
#pragma unroll
        for (int kb = 0; kb < NPOS; ++kb) sub_step(pe_hi[kb], pe_lo[kb]);
        // ---- fc_1 .. fc_4 (:103-106)
        for (int l = 1; l <= 4; ++l) plain_layer(l - 1, No(), cb_ + CB_BIAS + l * 256);
        // ---- fc_5 on cat([pos, x]) (:108): position FIRST
        {
            seam_half(First(), Yes(), No(), Full(), 4, cb_ + CB_BIAS + 5 * 256);
            unsigned a = frag + pipe.template acquire_as<SAVE>();
            seam_half(Second(), Yes(), No(), Full(), 4, cb_ + CB_BIAS + 5 * 256);
            mma_kblock<16, PIECES, 0, F2_IMAGE_BYTES>(acc, pe_hi[0], pe_lo[0], a, pipe);
            pipe.issue_done();
#pragma unroll
            for (int kb = 1; kb < NPOS; ++kb) sub_step(pe_hi[kb], pe_lo[kb]);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) sub_step(act_hi[kb], act_lo[kb]);
        }
        // ---- fc_6, fc_7 (:109-110)
        for (int l = 6; l <= 7; ++l) plain_layer(l - 1, No(), cb_ + CB_BIAS + l * 256);
        // ---- fc_8 (:113): rows 1..256 on the matrix pipe; the density row from the fp32 h7 in the seam
        plain_layer(7, Yes(), cb_ + CB_BIAS8);
        // ---- fc_9 on cat([x[:,1:], view_dir]) (:116-118), 128 rows: two k-blocks of 8-KiB images per sub-step; fc_8 has
        // no ReLU (:113)
        {
            seam_half(First(), No(), No(), Half(), 8, cb_ + CB_BIAS9);
            unsigned a = frag + pipe.template acquire_as<SAVE>();
            seam_half(Second(), No(), No(), Half(), 8, cb_ + CB_BIAS9);
            mma_kblock<8, PIECES, 0, F2_IMAGE_BYTES / 2>(acc, act_hi[0], act_lo[0], a, pipe);
            mma_kblock<8, 0, F2_IMAGE_BYTES, 3 * F2_IMAGE_BYTES / 2>(acc, act_hi[1], act_lo[1], a, pipe);
            pipe.issue_done();
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                a = frag + pipe.template acquire_as<SAVE>();
                mma_kblock<8, PIECES, 0, F2_IMAGE_BYTES / 2>(acc, act_hi[2 * j], act_lo[2 * j], a, pipe);
                mma_kblock<8, 0, F2_IMAGE_BYTES, 3 * F2_IMAGE_BYTES / 2>(acc, act_hi[2 * j + 1], act_lo[2 * j + 1], a, pipe);
                pipe.issue_done();
            }
            a = frag + pipe.template acquire_as<SAVE>();
            mma_kblock<8, PIECES, 0, F2_IMAGE_BYTES / 2>(acc, de_hi[0], de_lo[0], a, pipe);     // (NDIR = 1: + a zero k-block, skipped)
            if (NDIR > 1) mma_kblock<8, 0, F2_IMAGE_BYTES, 3 * F2_IMAGE_BYTES / 2>(acc, de_hi[NDIR - 1], de_lo[NDIR - 1], a, pipe);
            pipe.issue_done();
        }

        // ---- ReLU(fc_9), fc_out, sigmoid (:118-119) and sigma = relu(x[:,0]) (:115), fp32 vector ALU
        {
            const float unscale = cb_[F2_CB_UNSCALE + 9];
            float y[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) {
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = relu1(acc[fb][r] * unscale);
                if (SAVE) {
                    rec.store(pl_h9(MP), 128, fb, x, pipe);
                    rec.mask_bits(fb, x, mwords);
                    if (fb == 7) rec.store_mask(8, mwords, pipe);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(cb_ + CB_WOUT + c * HALF + 16 * fb + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[c] = fmaf(w[r], x[r], y[c]);
                }
            }
            float sp = sigma_pre + __shfl_xor(sigma_pre, 16, WAVE);
            sp += __shfl_xor(sp, 32, WAVE);
            amax = fmaxf(amax, __shfl_xor(amax, 16, WAVE));
            amax = fmaxf(amax, __shfl_xor(amax, 32, WAVE));
            const bool overflow = !(amax <= 65504.0f);      // an input, encoding or activation of this sample left the f16 range
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float p = y[c] + __shfl_xor(y[c], 16, WAVE);
                p += __shfl_xor(p, 32, WAVE);
                y[c] = 1.0f / (1.0f + expf(-(p + cb_[CB_SCALARS + 1 + c])));
            }
            if (valid && g == 0) {
                const float poison = overflow ? __builtin_nanf("") : 0.0f;    // loud, never a silent wrong value
                sigma_out[m] = fmaxf(sp + cb_[CB_SCALARS], 0.0f) + poison;
                rgb_out[3 * m + 0] = y[0] + poison;
                rgb_out[3 * m + 1] = y[1] + poison;
                rgb_out[3 * m + 2] = y[2] + poison;
            }
        }
    }
    if (!trailing && lag) pipe.idle_step();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---------------------------------------------------------------------------------------------------------------------
// The reverse chain (dX) on the split kernel: dY^T(l-1) = W_l^T dY^T(l), the fused family's mlp_bwd_dx_kernel (mlp_backward.hip)
// with split operands.  Same inputs (the forward's record: ReLU bit planes; sigma, rgb, their gradients), same outputs (the
// gradient planes dY0..dY8, dY9, dsig, gy in the workspace the dW kernels read; per-wavefront sums of the four scalar
// bias gradients).
// Gradients are not O(1) like activations: a sample's dY can sit at 1e-7, where f16 has no bits left.  Samples are the
// COLUMNS of these MFMAs, so every sample carries its own power-of-two scale: before a layer's dY is split it is multiplied
// by 2^t with t chosen from the sample's largest |dY| (-> [2^9, 2^10)), and the seam behind the contraction divides it out
// again (exact) before the plane store.  Within a column, elements below 2^-22 of its largest lose relative precision --
// they are below the fp32 rounding of the sums they enter.
// `plane_word` (0 = none): LDS byte address of the workgroup's running largest |dY| of this gradient plane, for the split-f16 dW
// GEMMs (mlp_backward.hip: they reduce OVER samples and take ONE scale per plane).  ds_max_u32 on the bit pattern of a
// non-negative float -- an LDS atomic (lgkmcnt), NOT a global one: one global atomic per wavefront and layer is 10 x M / 16
// returns to ten addresses of one L2 line (+3 ms at 1 M samples) and one more vmcnt event than SubPipe's counted waits know.
__device__ __forceinline__ float column_scale(float amax_lane, unsigned plane_word = 0) {
    float a = fmaxf(amax_lane, __shfl_xor(amax_lane, 16, WAVE));
    a = fmaxf(a, __shfl_xor(a, 32, WAVE));
    if (plane_word) {
        // lane group 0 holds the 16 samples' maxima; +inf / NaN (a poisoned forward) stay out: the scale falls back to 1
        if ((threadIdx.x & 48) == 0 && a < INFINITY)
            asm volatile("ds_max_u32 %0, %1" : : "v"(plane_word), "v"(__builtin_bit_cast(unsigned, a)) : "memory");
    }
    int e;
    (void)frexpf(a, &e);                       // a = f 2^e, f in [0.5, 1)
    int t = 10 - e;
    t = t > 100 ? 100 : (t < -100 ? -100 : t);
    return (a > 0.0f && a < INFINITY) ? ldexpf(1.0f, t) : 1.0f;
}

__global__ __launch_bounds__(64 * WAVES, 1) void mlp_bwd_dx_f16x2_kernel(const char *__restrict__ packed, int64_t M,
                                                                          const float *__restrict__ sigma,
                                                                          const float *__restrict__ rgb,
                                                                          const float *__restrict__ g_sigma,
                                                                          const float *__restrict__ g_rgb,
                                                                          const float *__restrict__ saved,
                                                                          float *__restrict__ dy,
                                                                          float *__restrict__ bias_partial,
                                                                          unsigned *__restrict__ plane_max) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    float *cb_ = reinterpret_cast<float *>(lds);
    const unsigned ring = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)CONST_BYTES;
    for (int e = tid; e < CONST_FLOATS / 4; e += 64 * WAVES) {
        f32x4 v = reinterpret_cast<const f32x4 *>(packed)[e];
        if (e >= F2_CB_PLANE_MAX / 4) v = f32x4{0.f, 0.f, 0.f, 0.f};      // the workgroup's plane maxima start at zero
        reinterpret_cast<f32x4 *>(cb_)[e] = v;
    }
    const unsigned frag = ring + (unsigned)f2_frag_offset(n, g);
    const unsigned plane_words = plane_max ? (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + 4u * F2_CB_PLANE_MAX : 0u;

    SubPipe pipe;
    pipe.src_wave = packed + CONST_BYTES + (int64_t)F2_SUBS * SUB_BYTES + wave * (PIECES * 1024);   // the transposed stream
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave = ring + (unsigned)wave * (PIECES * 1024u);
    pipe.issued = 0;
    pipe.issue_q = 0;
    pipe.consumed = 0;
    pipe.subs_per_tile = F2_BWD_SUBS;
    pipe.stores = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) pipe.issue_piece(p);
        pipe.issue_done();
    }
    const int64_t MP = padded_rows(M);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *masks = reinterpret_cast<const u32x4 *>(saved + pl_masks(MP));
    const int64_t ntiles = MP / TILE;
    float bsum4[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned gshift = 4u * (unsigned)(g >> 1);

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t m = tile * TILE + wave * WSAMPLES + n;
        const bool valid = m < M;
        const int64_t mc = valid ? m : M - 1;
        float gy[3], yv[3], gv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { yv[c] = rgb[3 * mc + c]; gv[c] = g_rgb[3 * mc + c]; }
        const float sg = sigma[mc], gsg = g_sigma[mc];
#pragma unroll
        for (int c = 0; c < 3; ++c) {       // fc_out + sigmoid (nerf.py:119): d y10 = g_rgb * rgb * (1 - rgb)
            const float t = gv[c] * yv[c] * (1.0f - yv[c]);
            gy[c] = valid ? t : 0.0f;
        }
        const float dsig = (valid && sg > 0.0f) ? gsg : 0.0f;      // sigma = relu(y8[0]) (:115)
        if (g == 0) {
            const f32x4 g4 = {gy[0], gy[1], gy[2], 0.0f};
            *reinterpret_cast<f32x4 *>(dy + gy_plane(MP) + 4 * m) = g4;
            dy[dsig_plane(MP) + m] = dsig;
            bsum4[0] += gy[0]; bsum4[1] += gy[1]; bsum4[2] += gy[2]; bsum4[3] += dsig;
        }
        pipe.stores += 2;

        Recorder rec = {};
        rec.open(dy, MP, m, g);
        f32x4 acc[16];
        f16x8 act_hi[8], act_lo[8];
        float dummy = 0.0f;
        // this lane's ReLU bits of plane `l`, shifted so that bit 16 ((fb >> 1) & 1) + 8 (fb & 1) + e of dword fb >> 2 is feature
        // 16 fb + 4 g + e (Recorder::mask_bits writes them at + 4 (g >> 1))
        auto load_mask = [&](int l) {
            u32x4 mk = masks[(int64_t)l * MP * 2 + 2 * m + (g & 1)];
#pragma unroll
            for (int d = 0; d < 4; ++d) mk[d] >>= gshift;
            return mk;
        };
        auto masked4 = [&](const u32x4 &mk, int fb, f32x4 v) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int keep;
                const int bit = 16 * ((fb >> 1) & 1) + 8 * (fb & 1) + e;
                switch (bit) {     // (v_bfe_i32 takes the offset as an immediate)
#define NERF_KEEP(B) case B: asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(mk[fb >> 2]), "n"(B)); break;
                    NERF_KEEP(0) NERF_KEEP(1) NERF_KEEP(2) NERF_KEEP(3) NERF_KEEP(8) NERF_KEEP(9) NERF_KEEP(10) NERF_KEEP(11)
                    NERF_KEEP(16) NERF_KEEP(17) NERF_KEEP(18) NERF_KEEP(19) NERF_KEEP(24) NERF_KEEP(25) NERF_KEEP(26) NERF_KEEP(27)
#undef NERF_KEEP
                    default: keep = 0;
                }
                // (through a scalar: __builtin_bit_cast applied to the element expression v[e] itself reads element 0 of the
                // vector whatever e is -- clang 19 / ROCm 7.2 -- and every lane's e = 1..3 came out as keep_e & keep_0 & v[0])
                const float t = v[e];
                v[e] = __builtin_bit_cast(float, __builtin_bit_cast(int, t) & keep);
            }
            return v;
        };
        // NB blocks of dY held in acc[] (true values) -> packed hi / lo of dY * 2^t; returns 2^-t
        auto scale_and_split = [&](auto nb_tag, int plane) {
            constexpr int NB = decltype(nb_tag)::value;
            float amax = 0.0f;
#pragma unroll
            for (int fb = 0; fb < NB; ++fb) {
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(acc[fb][0]), "v"(acc[fb][1]));
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(acc[fb][2]), "v"(acc[fb][3]));
            }
            const float cs = column_scale(amax, plane_words ? plane_words + 4u * (unsigned)plane : 0u);
#pragma unroll
            for (int fb = 0; fb < NB; ++fb) {
                const f32x4 x = acc[fb] * cs;
                split4(x, act_hi[fb >> 1], act_lo[fb >> 1], fb & 1, dummy);
            }
            return 1.0f / cs;
        };
        typedef std::integral_constant<int, 8> Eight;
        typedef std::integral_constant<int, 16> Sixteen;

        // ---- dY9 = (W_out^T d y10) . [h9 > 0]   (vector ALU, 3 x 128 MACs per sample)
        u32x4 mk = load_mask(8);
#pragma unroll
        for (int fb = 0; fb < 8; ++fb) {
            const int k0 = 16 * fb + 4 * g;
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(cb_ + CB_WOUT + k0);
            const f32x4 w1 = *reinterpret_cast<const f32x4 *>(cb_ + CB_WOUT + HALF + k0);
            const f32x4 w2 = *reinterpret_cast<const f32x4 *>(cb_ + CB_WOUT + 2 * HALF + k0);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(w2[e], gy[2], fmaf(w1[e], gy[1], w0[e] * gy[0]));
            acc[fb] = masked4(mk, fb, v);
            rec.store(dy9_plane(MP), 128, fb, acc[fb], pipe);
        }
        float inv_cs = scale_and_split(Eight(), 9);
        u32x4 mk_next = load_mask(7);
        // ---- d y8[1:257] = W9[:, 0:256]^T dY9: four sub-steps, accumulators from zero
#pragma unroll
        for (int fb = 0; fb < 16; ++fb) acc[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const unsigned a = frag + pipe.template acquire_as<true>();
            mma_kblock<16, PIECES, 0, F2_IMAGE_BYTES>(acc, act_hi[kb], act_lo[kb], a, pipe);
            pipe.issue_done();
        }
        // ---- l = 8 .. 1: dY(l) = [previous contraction] . [h_l > 0] (fc_8 has no ReLU), stored; dY(l-1) pre-mask = W_l^T dY(l)
        for (int l = 8; l >= 1; --l) {
            const float un = cb_[F2_CB_UNSCALE + (l == 8 ? 9 : l + 1)] * inv_cs;
            mk = mk_next;
#pragma unroll
            for (int fb = 0; fb < 16; ++fb) {
                f32x4 x = acc[fb] * un;
                if (l < 8) x = masked4(mk, fb, x);
                acc[fb] = x;
                rec.store(dy_plane(MP, l), 256, fb, x, pipe);
            }
            if (l > 1) mk_next = load_mask(l == 8 ? 7 : l - 1);      // (l = 8 uses none: h7's bits serve the seam of l = 7)
            else mk_next = load_mask(0);
            inv_cs = scale_and_split(Sixteen(), l);
            if (l == 8) {      // the density row of fc_8 contributes w8[0, k] * d y8[0]: into the (scaled) accumulators
                const float init = dsig * cb_[F2_CB_SCALE + 8] / inv_cs;
#pragma unroll
                for (int fb = 0; fb < 16; ++fb) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(cb_ + CB_W8ROW0 + 16 * fb + 4 * g);
                    acc[fb] = wv * init;
                }
            } else {
#pragma unroll
                for (int fb = 0; fb < 16; ++fb) acc[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const unsigned a = frag + pipe.template acquire_as<true>();
                mma_kblock<16, PIECES, 0, F2_IMAGE_BYTES>(acc, act_hi[kb], act_lo[kb], a, pipe);
                pipe.issue_done();
            }
        }
        // ---- dY0: mask with h0 and store (the encodings carry no gradient)
        {
            const float un = cb_[F2_CB_UNSCALE + 1] * inv_cs;
            float amax0 = 0.0f;
#pragma unroll
            for (int fb = 0; fb < 16; ++fb) {
                const f32x4 x = masked4(mk_next, fb, acc[fb] * un);
                rec.store(dy_plane(MP, 0), 256, fb, x, pipe);
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax0) : "v"(x[0]), "v"(x[1]));
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax0) : "v"(x[2]), "v"(x[3]));
            }
            if (plane_words) (void)column_scale(amax0, plane_words);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = bsum4[c];       // lanes of the other lane groups hold zeros
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
        if (lane == 0) bias_partial[((int64_t)blockIdx.x * WAVES + wave) * 4 + c] = v;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (plane_max) {      // the workgroup's ten maxima into the launch's (order-independent)
        __syncthreads();
        if (tid < 10) {
            const unsigned w = reinterpret_cast<const unsigned *>(cb_)[F2_CB_PLANE_MAX + tid];
            if (w) atomicMax(plane_max + tid, w);
        }
    }
}

}  // namespace

NERF_API int nerf_mlp_forward_f16x2(const nerf_net_t *net_abi, const void *packed_f16x2, const float *pos,
                                    const float *view_dir, int64_t M, float *sigma, float *rgb, nerf_stream_t stream) {
    mlp::Net net;
    if (int rc = nerf::f16x2_net(net_abi, net, "nerf_mlp_forward_f16x2")) return rc;
    if (!nerf::raw_inputs_ok(net))
        return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_mlp_forward_f16x2: the kernel encodes raw points: nerf_net_t needs the "
                                                "levels of both PositionalEncoders");
    NERF_REQUIRE(M >= 0, "nerf_mlp_forward_f16x2: negative M");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(packed_f16x2 && pos && view_dir && sigma && rgb, "nerf_mlp_forward_f16x2: null pointer");
    const mlp::F2Layout L = mlp::f2_layout(net.e_pos, net.e_dir);
    typedef void (*Kernel)(const Net, const char *, const float *, const float *, int64_t, float *, float *, float *);
    static const Kernel kernels[3][2] = {{mlp_forward_f16x2_kernel<2, 1, false>, mlp_forward_f16x2_kernel<2, 2, false>},
                                         {mlp_forward_f16x2_kernel<3, 1, false>, mlp_forward_f16x2_kernel<3, 2, false>},
                                         {mlp_forward_f16x2_kernel<4, 1, false>, mlp_forward_f16x2_kernel<4, 2, false>}};
    static nerf::DeviceMask configured[3][2] = {{{0}, {0}}, {{0}, {0}}, {{0}, {0}}};
    const Kernel kern = kernels[L.npos - 2][L.ndir - 1];
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), F2_LDS_BYTES, configured[L.npos - 2][L.ndir - 1],
                                          "nerf_mlp_forward_f16x2: LDS attribute"))
        return rc;
    const int cus = nerf::device_cus();
    const int64_t ntiles = (M + TILE - 1) / TILE;
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(64 * WAVES), F2_LDS_BYTES,
                       nerf::as_stream(stream), net, static_cast<const char *>(packed_f16x2), pos, view_dir, M, sigma, rgb,
                       static_cast<float *>(nullptr));
    return nerf::check_launch("nerf_mlp_forward_f16x2");
}

// The TRAINING forward on the split kernel: same outputs, plus the activation record of nerf_mlp_forward(saved != NULL)
// (nerf_mlp_saved_bytes), which nerf_mlp_backward reads.  Fused family behind PositionalEncoders, raw points.
NERF_API int nerf_mlp_forward_f16x2_record(const nerf_net_t *net_abi, const void *packed_f16x2, const float *pos,
                                           const float *view_dir, int64_t M, float *sigma, float *rgb, void *saved,
                                           nerf_stream_t stream) {
    mlp::Net net;
    if (int rc = nerf::fused_net(net_abi, net, "nerf_mlp_forward_f16x2_record")) return rc;
    if (!nerf::raw_inputs_ok(net))
        return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_mlp_forward_f16x2_record: the kernel encodes raw points: nerf_net_t needs "
                                                "the levels of both PositionalEncoders");
    NERF_REQUIRE(M >= 0, "nerf_mlp_forward_f16x2_record: negative M");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(packed_f16x2 && pos && view_dir && sigma && rgb && saved, "nerf_mlp_forward_f16x2_record: null pointer");
    auto kern = mlp_forward_f16x2_kernel<2, 1, true>;
    static nerf::DeviceMask configured = {0};
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), F2_LDS_BYTES, configured,
                                          "nerf_mlp_forward_f16x2_record: LDS attribute"))
        return rc;
    const int cus = nerf::device_cus();
    const int64_t ntiles = (M + TILE - 1) / TILE;
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(64 * WAVES), F2_LDS_BYTES,
                       nerf::as_stream(stream), net, static_cast<const char *>(packed_f16x2), pos, view_dir, M, sigma, rgb,
                       static_cast<float *>(saved));
    return nerf::check_launch("nerf_mlp_forward_f16x2_record");
}

// mlp_backward.hip calls this for stage 1 when it is handed a split-f16 stream (nerf_mlp_backward_f16x2): the reverse chain
// on the f16 matrix pipe; `partials` = number of per-wavefront bias partials written (the reducer's loop bound)
namespace nerf {
int launch_dx_f16x2(const void *packed_f16x2, int64_t M, const float *sigma, const float *rgb, const float *g_sigma,
                    const float *g_rgb, const float *saved, float *dy, float *bias_partial, int *partials, unsigned *plane_max,
                    hipStream_t s) {
    static nerf::DeviceMask configured = {0};
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_bwd_dx_f16x2_kernel), F2_LDS_BYTES, configured,
                                          "nerf_mlp_backward_f16x2: LDS attribute (dX)"))
        return rc;
    const int cus = nerf::device_cus();
    const int64_t ntiles = mlp::padded_rows(M) / TILE;
    const unsigned grid = (unsigned)(ntiles < cus ? ntiles : (cus < 512 ? cus : 512));
    hipLaunchKernelGGL(mlp_bwd_dx_f16x2_kernel, dim3(grid), dim3(64 * WAVES), F2_LDS_BYTES, s,
                       static_cast<const char *>(packed_f16x2), M, sigma, rgb, g_sigma, g_rgb, saved, dy, bias_partial, plane_max);
    *partials = (int)grid * WAVES;
    return nerf::check_launch("nerf_mlp_backward_f16x2: dx chain");
}
}  // namespace nerf
