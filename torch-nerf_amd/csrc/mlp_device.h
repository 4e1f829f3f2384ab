// Device-side building blocks shared by the fused MLP kernels (forward, backward dX chain, dW).
// See mlp_layout.h for the fragment algebra and the packed stream these operate on.
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace mlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// sin and cos with a 3-term Cody-Waite reduction (FMA) and Cephes minimax polynomials:
// <= ~1.5e-7 abs error for |x| < 3e4.  Branch-free so that the compiler can interleave the 48
// independent evaluations of a tile; EXACT selects the library path (correct for any argument).
template <bool EXACT>
__device__ __forceinline__ void sincos_cw(float x, float &s, float &c) {
    if (EXACT) {
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

// feature k of PositionalEncoder(3, L, include_input).encode((x,y,z)); 0 beyond kmax = its out_dim
// layout (positional_encoder.py:83-88): [x y z (if include_input) | sin(2^0 xyz) cos(2^0 xyz) | sin(2^1 xyz) ...]
template <bool EXACT>
__device__ __forceinline__ float enc_feature(int k, float x, float y, float z, int kmax, int include_input = 1) {
    const int raw = include_input ? 3 : 0;
    const int e = k - raw;
    const int f = e < 0 ? 0 : e / 6;
    const int r6 = e - 6 * f;
    const int ch = k < raw ? k : (r6 >= 3 ? r6 - 3 : r6);
    const float v = ch == 0 ? x : (ch == 1 ? y : z);
    float s, c;
    sincos_cw<EXACT>(ldexpf(v, f), s, c);
    const float t = r6 >= 3 ? c : s;
    return k < raw ? v : (k < kmax ? t : 0.0f);
}

// All NF (64 | 32) encoding features of one sample, tail beyond 3 + 6 LEVELS zero: ONE sincos per (octave, channel)
// gives both its sin and its cos feature -- the same sincos_cw<false>(ldexpf(v, f)) that enc_feature evaluates, so the
// values are bit-identical to it.  Every index is a compile-time constant.  (fp32 MFMAs and vector instructions share
// the SIMD lanes -- DESIGN.md section 4.2 -- so the encodings are paid in full: enc_feature per needed feature costs
// 48 evaluations + run-time index arithmetic per lane and tile, ~1.6 k instructions; this table + the per-half select
// below ~1.1 k.)
template <int LEVELS, int NF>
__device__ __forceinline__ void encode_table(float x, float y, float z, float (&F)[NF]) {
#pragma unroll
    for (int k = 0; k < NF; ++k) F[k] = 0.0f;
    const float v[3] = {x, y, z};
#pragma unroll
    for (int c = 0; c < 3; ++c) F[c] = v[c];
#pragma unroll
    for (int f = 0; f < LEVELS; ++f)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s, co;
            sincos_cw<false>(ldexpf(v[c], f), s, co);
            F[3 + 6 * f + c] = s;
            F[3 + 6 * f + 3 + c] = co;
        }
}

// B-fragment registers of encoding block `blk` for lane half h: register r <-> feature 32 blk + (r&3) + 8 (r>>2) + 4 h
template <int NF>
__device__ __forceinline__ void table_to_fragment(const float (&F)[NF], int blk, int h, f32x16 &frag) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = 32 * blk + (r & 3) + 8 * (r >> 2);
        frag[r] = h ? F[k + 4] : F[k];
    }
}

// largest |argument| the encodings of this sample will see: 2^(L-1) * max|coordinate|
__device__ __forceinline__ bool encoding_needs_exact(const float (&raw)[6], int l_pos, int l_dir) {
    const float p = fmaxf(fmaxf(fabsf(raw[0]), fabsf(raw[1])), fabsf(raw[2]));
    const float d = fmaxf(fmaxf(fabsf(raw[3]), fabsf(raw[4])), fabsf(raw[5]));
    return !(ldexpf(p, l_pos - 1) < 30000.0f && ldexpf(d, l_dir - 1) < 30000.0f);  // also true for NaN
}

// ReLU as ONE instruction.  fmaxf(x, 0) on a raw MFMA result compiles to two v_max_f32 (hipcc first
// canonicalises a value it cannot prove quiet; it folds v_med3(x, 0, inf) back into the same pair), and every
// vector instruction of a layer seam is exposed: one wavefront per SIMD has no other wave to hide it behind.
// v_max_f32 itself needs no canonical input (IEEE mode quiets NaNs: NaN -> 0, -0 -> +0, like fmaxf).
__device__ __forceinline__ float relu1(float x) {
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
}

// one 1-KiB piece per instruction, wave-uniform base + per-lane byte offset:
// LDS[m0 + lane*16] <- global[src + lane_off].  The address arithmetic of a piece is then scalar (the
// per-lane form costs a 64-bit vector add per piece, in issue slots the MFMA stream cannot hide) and M0 is
// simply overwritten.  It cannot be declared: M0 is a RESERVED register for hipcc, which answers an "m0" clobber
// with "reserved registers on the clobber list may not be preserved" (once per inlined piece) and keeps its own
// hoisted M0 initialisations regardless.  So the contract is enforced from outside: gfx9+ LDS instructions do
// not need M0, these kernels use nothing else that reads it (no s_movrel, no v_readlane with M0, no GWS), and
// tests/test_isa_audit.py fails the CPU test suite if the emitted ISA of any MLP kernel ever reads M0 outside
// a piece.
#ifndef X_DMA_POLICY
#define X_DMA_POLICY ""      // A/B knob: cache-policy suffix of the weight-stream loads (" nt", " sc0", " sc1")
#endif
__device__ __forceinline__ void lds_dma_16s(const char *src, unsigned lane_off, unsigned lds_dst) {
#ifdef X_DMA_BUILTIN   // A/B: the compiler's own LDS-DMA -- it sets M0 and counts vmcnt itself
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane_off),
                                     (__attribute__((address_space(3))) void *)(uintptr_t)lds_dst, 16, 0, 0);
    return;
#endif
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1" X_DMA_POLICY
        :
        : "v"(lane_off), "s"(src), "s"(lds_dst)
        : "memory");
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

// Weight stream -> LDS ring, one PAIR of 32-KiB chunks per step (= 256 MFMAs per wavefront).
// The ring holds two pairs: while pair p is consumed, pair p+1 is in flight.  Each wave copies its
// quarter of every chunk (8 one-KiB pieces); completion is per wave (vmcnt) + one workgroup barrier.
struct Pipe {
    const char *src_wave;  // stream base + this wave's byte offset inside a chunk (wave-uniform)
    unsigned lane_off;     // lane * 16
    unsigned lds_wave;     // LDS byte address of ring slot 0 + this wave's offset
    unsigned issued;       // pairs issued so far
    int issue_pos;         // stream position (in pairs) of the next pair to issue
    unsigned consumed;     // pairs consumed so far
    int n_pairs;
    unsigned long long skip_mask;   // bit p set: the walk jumps over stream position p (issue_pos starts at the first
                           // position that is not skipped) -- the fc_9
                           // direction pair when the caller supplies the per-ray direction contribution itself
                           // (render_fused.hip), the input-gradient pairs of the transposed stream when the dX chain
                           // is not asked for them (mlp_backward.hip)

    // one of the 16 one-KiB pieces this wave copies per pair (piece 0..7 -> chunk 0, 8..15 -> chunk 1)
    __device__ __forceinline__ void issue_piece(int piece) const {
        const int c = piece >> 3, j = piece & 7;
        lds_dma_16s(src_wave + (size_t)issue_pos * PAIR_BYTES + c * CHUNK_BYTES + j * 1024, lane_off,
                    lds_wave + (issued & 1) * PAIR_BYTES + c * CHUNK_BYTES + j * 1024);
    }
    __device__ __forceinline__ void issue_done() {
        ++issued;
        do {
            issue_pos = (issue_pos + 1 == n_pairs) ? 0 : issue_pos + 1;
        } while (issue_pos < 64 && ((skip_mask >> issue_pos) & 1ull));   // (positions >= 64 cannot be skipped: a 64-bit mask)
    }
    __device__ __forceinline__ void issue() {
#pragma unroll
        for (int p = 0; p < 16; ++p) issue_piece(p);
        issue_done();
    }
    // Make the next pair readable; returns the LDS byte offset (from slot 0) of its first chunk.
    // Only the pair being acquired is outstanding at this point, so a plain vmcnt(0) is exact --
    // and stays exact whatever other loads/stores the wave has queued.  The copy of the FOLLOWING
    // pair is not issued here: mma_pair() spreads its 16 DMA instructions between the MFMAs of
    // this pair, where they issue for free behind the matrix pipe.
    __device__ __forceinline__ unsigned acquire() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's pieces landed; everyone left the pair refilled next
        asm volatile("" ::: "memory");
        const unsigned off = (consumed & 1) * PAIR_BYTES;
        ++consumed;
        return off;
    }
};

// acc[fb] += W[32 fb .. 32 fb + 31][32 k-values of this chunk] . b   for NFB feature blocks.
// If N_PIECES > 0, the wave also issues LDS-DMA pieces FIRST_PIECE .. FIRST_PIECE + N_PIECES - 1 of
// the next pair, evenly interleaved with the MFMA groups of this chunk.
// LDS -> 4 VGPRs, issued by hand so that hipcc cannot sink it back to just before its use (it does,
// whatever sched_group_barrier says, and then reuses ONE fragment buffer for the whole kernel).
// The destination is NOT protected by the compiler's waitcnt bookkeeping: every use must sit
// behind lds_fragments_ready().
__device__ __forceinline__ f32x4 lds_read_fragment(unsigned lds_addr, int imm_offset) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(imm_offset));
    return v;
}
__device__ __forceinline__ void lds_fragments_ready() {
    __builtin_amdgcn_sched_barrier(0);  // the previous group's MFMAs stay above the wait (they cover it) ...
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);  // ... and the consumers of the fragment stay below it
}

// A plane store spread over the MFMA stream (record forward, dX chain): the 4 NFB one-KiB stores of save_plane(),
// issued one at a time between MFMA groups instead of back to back at the layer seam.  A wavefront gets a 1-KiB store
// out every ~190 cycles (four wavefronts share the CU's ~16 B / clock store path: scripts/timeline_layered.py), and at
// the seam nothing hides that: 32 stores = 6 k cycles of a 65 k-cycle layer.  The blocks stored must stay untouched
// until the last store has been issued (they are the B operands of the layer being multiplied: they do).
// (The data operand is a VGPR: where the allocator keeps some of the stored blocks in AGPRs -- the layered family's
// register-resident record forward -- every store needs a four-register copy; an "a" operand moves the problem to the
// blocks that sit in VGPRs.  Those kernels keep their stores at the seam.)
struct PlaneStore {
    uint64_t tile;         // address of this wavefront's 32-sample tile of the plane (wave-uniform: SGPR pair)
    unsigned unit16;       // (2 i + h) << 4
    const f32x16 *blk;
    __device__ __forceinline__ void open(float *plane, int width, int64_t m, int h, const f32x16 *blocks) {
        const int i = (int)(m & 31);
        const uint64_t a = reinterpret_cast<uint64_t>(plane + (m - i) * width);
        tile = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)a);
        unit16 = (unsigned)(2 * i + h) << 4;
        blk = blocks;
    }
    // store s = 4 fb + q (the order save_plane() walks); `s` is a constant after unrolling
    __device__ __forceinline__ void issue(int s) const {
        const int fb = s >> 2, q = s & 3;
        const f32x4 v = {blk[fb][4 * q], blk[fb][4 * q + 1], blk[fb][4 * q + 2], blk[fb][4 * q + 3]};
#ifdef X_STORE_PLAIN   // A/B: a store hipcc can see (its own hazard recognizer, its own address form)
        *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(tile + (uint64_t)fb * 4096u) + (unit16 ^ (32u * q)) + q * 1024) = v;
        return;
#endif
        // (s_nop 4 in front: the scalar base may come straight out of a v_readfirstlane / SALU add -- see save_plane();
        // s_nop 1 behind: wide-store data hazard, inside the statement)
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1"
                     : : "v"(unit16 ^ (32u * q)), "v"(v), "s"(tile + (uint64_t)fb * 4096u), "n"(q * 1024) : "memory");
    }
};

// N_STORES > 0: stores FIRST_STORE .. FIRST_STORE + N_STORES - 1 of `st` are issued between the groups of this chunk too.
template <int NFB, int FIRST_PIECE = 0, int N_PIECES = 0, bool FRESH = false, int N_STORES = 0>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[8], const f32x16 &b, const char *chunk,
                                          const int (&offq)[4], const Pipe *pipe = nullptr, const PlaneStore *st = nullptr,
                                          int first_store = 0) {
    constexpr int GROUPS = 4 * NFB, EVERY = GROUPS / (N_PIECES > 0 ? N_PIECES : GROUPS);
    constexpr int SEVERY = GROUPS / (N_STORES > 0 ? N_STORES : GROUPS);
    static_assert(N_STORES == 0 || GROUPS % N_STORES == 0, "stores spread evenly over the groups");
    // A fragments are fetched one (q, fb) group ahead of the MFMAs that consume them, into two
    // alternating buffers: a ds_read_b128 issued behind a group's last MFMA returns ~80 cycles
    // later than the matrix pipe frees up, which with a single buffer costs ~14 idle cycles per group.
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)chunk;
    unsigned addr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) addr[q] = base + (unsigned)offq[q];
    f32x4 abuf[2];
    abuf[0] = lds_read_fragment(addr[0], 0);
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const int q = g / NFB, fb = g % NFB;
        lds_fragments_ready();  // abuf[g & 1] has landed
        if (g + 1 < GROUPS) abuf[(g + 1) & 1] = lds_read_fragment(addr[(g + 1) / NFB], ((g + 1) % NFB) * 4096);
        const f32x4 a = abuf[g & 1];
        if (FRESH && q == 0) {   // the accumulator block starts here: C = 0 instead of 16 zeroing writes
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * q + 0], zero, 0, 0, 0);
        } else {
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * q + 0], acc[fb], 0, 0, 0);
        }
        acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * q + 1], acc[fb], 0, 0, 0);
        if (N_PIECES > 0 && g % EVERY == 0) pipe->issue_piece(FIRST_PIECE + g / EVERY);
        acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * q + 2], acc[fb], 0, 0, 0);
        if (N_STORES > 0 && g % SEVERY == SEVERY - 1) st->issue(first_store + g / SEVERY);
        acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * q + 3], acc[fb], 0, 0, 0);
    }
}

// The same for a SLOT-MAJOR chunk (mlp_layout.h, input-gradient pairs of the transposed stream): eight 4-KiB slots, slot
// kb * NFB + fb = rows 32 fb .. 32 fb + 31 of the (thin) output x the 32 k-values of operand b[kb], NKB * NFB <= 8:
//   acc[fb] += sum over kb of W_slot(kb, fb) . b[kb]
struct NoHook { __device__ __forceinline__ void operator()(int, int) const {} };
// `hook(g, GROUPS)` runs once in every MFMA group (the layered kernel's operand loads for the NEXT pair ride there: a
// wavefront gets a 1-KiB vector-memory instruction out every 60..190 cycles, and in front of the MFMAs that is exposed).
template <int NFB, int NKB, int FIRST_PIECE = 0, int N_PIECES = 0, bool FRESH = false, class Hook = NoHook>
__device__ __forceinline__ void mma_slots(f32x16 *acc, const f32x16 *b, const char *chunk, const int (&offq)[4],
                                          const Pipe *pipe = nullptr, Hook hook = Hook()) {
    static_assert(NFB * NKB <= 8, "a chunk holds eight slots");
    constexpr int GROUPS = 4 * NFB * NKB, EVERY = GROUPS / (N_PIECES > 0 ? N_PIECES : GROUPS);
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)chunk;
    unsigned addr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) addr[q] = base + (unsigned)offq[q];
    f32x4 abuf[2];
    abuf[0] = lds_read_fragment(addr[0], 0);
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const int kb = g / (4 * NFB), q = (g / NFB) % 4, fb = g % NFB;
        lds_fragments_ready();
        if (g + 1 < GROUPS) {
            const int g1 = g + 1, kb1 = g1 / (4 * NFB), q1 = (g1 / NFB) % 4, fb1 = g1 % NFB;
            abuf[g1 & 1] = lds_read_fragment(addr[q1], (kb1 * NFB + fb1) * 4096);
        }
        const f32x4 a = abuf[g & 1];
        if (FRESH && kb == 0 && q == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[kb][4 * q + 0], zero, 0, 0, 0);
        } else {
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[kb][4 * q + 0], acc[fb], 0, 0, 0);
        }
        acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[kb][4 * q + 1], acc[fb], 0, 0, 0);
        if (N_PIECES > 0 && g % EVERY == 0 && g / EVERY < N_PIECES) pipe->issue_piece(FIRST_PIECE + g / EVERY);
        acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[kb][4 * q + 2], acc[fb], 0, 0, 0);
        hook(g, GROUPS);
        acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[kb][4 * q + 3], acc[fb], 0, 0, 0);
    }
}

// Two SAMPLE blocks per wavefront (mlp_layered.hip, narrow networks: 64 samples x 128 features fill the registers that
// 32 samples x 256 features fill in the wide kernels): every A fragment feeds 8 MFMAs instead of 4 -- half the LDS
// reads and half the weight stream per MFMA.  Slot-major chunk as in mma_slots, slot = kb * STRIDE + fb; the k-blocks
// multiplied are kb = 0 .. NKB-1, the operands of k-block kb are bsel(0, kb) / bsel(1, kb) (references to f32x16).
// N_STORES > 0: plane stores first_store .. first_store + N_STORES - 1 (those below S_TOTAL) ride between the groups as
// well (PlaneStore): store s belongs to sample block s / SPS (st[s / SPS]) and is its store s % SPS.  `first_store` is a
// constant after unrolling.
template <int NFB, int NKB, int STRIDE, int FIRST_PIECE = 0, int N_PIECES = 0, bool FRESH = false, int NSB = 2,
          int N_STORES = 0, int SPS = 1, int S_TOTAL = 0, class BSel>
__device__ __forceinline__ void mma_slots2(f32x16 *acc0, f32x16 *acc1, BSel bsel, const char *chunk, const int (&offq)[4],
                                           const Pipe *pipe = nullptr, const PlaneStore *st = nullptr, int first_store = 0) {
    static_assert(NFB <= STRIDE && STRIDE * NKB <= 8, "a chunk holds eight slots");
    constexpr int GROUPS = 4 * NFB * NKB;
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)chunk;
    unsigned addr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) addr[q] = base + (unsigned)offq[q];
    f32x4 abuf[2];
    abuf[0] = lds_read_fragment(addr[0], 0);
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const int kb = g / (4 * NFB), q = (g / NFB) % 4, fb = g % NFB;
        lds_fragments_ready();
        if (g + 1 < GROUPS) {
            const int g1 = g + 1, kb1 = g1 / (4 * NFB), q1 = (g1 / NFB) % 4, fb1 = g1 % NFB;
            abuf[g1 & 1] = lds_read_fragment(addr[q1], (kb1 * STRIDE + fb1) * 4096);
        }
        const f32x4 a = abuf[g & 1];
        const f32x16 &b0 = bsel(0, kb), &b1 = bsel(NSB - 1, kb);      // (NSB = 1: one sample block, acc1 / b1 unused)
        if (FRESH && kb == 0 && q == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc0[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0[4 * q + 0], zero, 0, 0, 0);
            if (NSB > 1) acc1[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1[4 * q + 0], zero, 0, 0, 0);
        } else {
            acc0[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0[4 * q + 0], acc0[fb], 0, 0, 0);
            if (NSB > 1) acc1[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1[4 * q + 0], acc1[fb], 0, 0, 0);
        }
        acc0[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0[4 * q + 1], acc0[fb], 0, 0, 0);
        if (NSB > 1) acc1[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1[4 * q + 1], acc1[fb], 0, 0, 0);
        if (N_PIECES > 0) {   // the next pair's DMA pieces, spread evenly over this chunk's groups
#pragma unroll
            for (int pp = g * N_PIECES / GROUPS; pp < (g + 1) * N_PIECES / GROUPS; ++pp) pipe->issue_piece(FIRST_PIECE + pp);
        }
        acc0[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0[4 * q + 2], acc0[fb], 0, 0, 0);
        if (NSB > 1) acc1[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1[4 * q + 2], acc1[fb], 0, 0, 0);
        if (N_STORES > 0) {
#pragma unroll
            for (int ss = g * N_STORES / GROUPS; ss < (g + 1) * N_STORES / GROUPS; ++ss)
                if (first_store + ss < S_TOTAL) st[(first_store + ss) / SPS].issue((first_store + ss) % SPS);
        }
        acc0[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0[4 * q + 3], acc0[fb], 0, 0, 0);
        if (NSB > 1) acc1[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1[4 * q + 3], acc1[fb], 0, 0, 0);
    }
}

// One pipeline step: both chunks of the acquired pair.  The next pair's 16 DMA pieces are all
// issued during the FIRST chunk, so the youngest of them still has a whole chunk of MFMAs
// (8 k cycles) to land before the next acquire waits for it.
// N_STORES > 0: that many stores of `st`, from `first_store` on, ride in the first chunk as well (they too have the
// second chunk to complete before the next acquire's vmcnt(0)).
template <int NFB, bool FRESH = false, int N_STORES = 0>
__device__ __forceinline__ void mma_pair(f32x16 (&acc)[8], const f32x16 &b0, const f32x16 &b1, const char *w,
                                         const int (&offq)[4], Pipe &pipe, const PlaneStore *st = nullptr,
                                         int first_store = 0) {
    mma_chunk<NFB, 0, 16, FRESH, N_STORES>(acc, b0, w, offq, &pipe, st, first_store);
    mma_chunk<NFB>(acc, b1, w + CHUNK_BYTES, offq);
    pipe.issue_done();
}

// x = act(raw accumulator block + bias); `bias_blk` points at the 32 biases of the block
template <bool RELU>
__device__ __forceinline__ f32x16 finish_block(const f32x16 &raw, const float *bias_blk, int h) {
    f32x16 x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_blk + 8 * q + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = raw[4 * q + j] + bv[j];
            x[4 * q + j] = RELU ? relu1(v) : v;
        }
    }
    return x;
}

// this lane's half of dot(w[32 features of a block], x); w in LDS
__device__ __forceinline__ float block_dot(const float *w_blk, const f32x16 &x, int h) {
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w_blk + 8 * q + 4 * h);
        s = fmaf(v.x, x[4 * q + 0], s);
        s = fmaf(v.y, x[4 * q + 1], s);
        s = fmaf(v.z, x[4 * q + 2], s);
        s = fmaf(v.w, x[4 * q + 3], s);
    }
    return s;
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

// Sample m of a TF-layout plane (mlp_layout.h) <- this lane's 4-feature groups: one fully coalesced 1-KiB
// store per (fb, q) slot.  m - (m & 31) is the same in all lanes of the wavefront, so the tile address is
// scalar and the per-lane part is four 32-bit offsets.  Stores are unconditional: rows in [M, MP) receive the
// (finite) values of the clamped lane.  The backward GEMMs stay exact because the GRADIENT planes hold exact
// zeros there (upstream gradients of padded lanes are zero), so a padded row contributes 0 * finite = 0 to
// every sum.
// PAD: every store is followed by the two wait states gfx950 wants between a wide VMEM store and a VALU write of its
// data registers -- hipcc pads its own stores but cannot see these; kernels whose register allocation is tight enough
// to reuse a data register at once (the input-gradient dX chain: found by scripts/audit_asm_loads.py) ask for it.
template <int NFB, bool PAD = false>
__device__ __forceinline__ void save_plane(float *plane, int width, int64_t m, int h, const f32x16 *blk) {
    const int i = (int)(m & 31);
    const uint64_t tile_addr = reinterpret_cast<uint64_t>(plane + (m - i) * width);
    // The tile address is wave-uniform: it goes into an SGPR pair and every store is
    //     global_store_dwordx4 v_lane_offset, v[data], s[base:base+1] offset:q*1024
    // with the feature block stepping the scalar base (SALU) and ONE lane register, (2 i + h) << 4, XORed with
    // 32 q per slot.  Left to hipcc the stores take a 64-bit vector address each: the four lane offsets end up parked
    // in AGPRs and every store costs two v_accvgpr_read + a 64-bit add, in a seam where every vector instruction is
    // exposed.
    const uint64_t tile = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(tile_addr >> 32)) << 32) |
                          (unsigned)__builtin_amdgcn_readfirstlane((int)tile_addr);
    const unsigned unit16 = (unsigned)(2 * i + h) << 4;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
        const uint64_t base = tile + (uint64_t)fb * 4096u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {blk[fb][4 * q], blk[fb][4 * q + 1], blk[fb][4 * q + 2], blk[fb][4 * q + 3]};
#ifdef X_STORE_PLAIN
            *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(base) + (unit16 ^ (32u * q)) + q * 1024) = v;
            continue;
#endif
            // The scalar base comes out of v_readfirstlane (a VALU write of an SGPR), and gfx950 wants 5 wait states
            // between that and a VMEM instruction reading the SGPR.  hipcc's hazard recognizer cannot see the VMEM
            // instruction inside the asm, so the FIRST store of the call carries the wait states itself (measured
            // without them: in the pre-encoded forward, where nothing else sits between the two, one store in ~60
            // went to the previous plane's address -- stale PE / DE planes, 5 % gradient error, run to run different;
            // scripts/audit_asm_loads.py now checks every asm VMEM instruction for this).
            // (PAD: the wait states ride in the SAME asm statement -- a separate one may be scheduled away from its store)
            if (fb == 0 && q == 0) {
                if (PAD) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1"
                                      : : "v"(unit16 ^ (32u * q)), "v"(v), "s"(base), "n"(q * 1024) : "memory");
                else asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3"
                                  : : "v"(unit16 ^ (32u * q)), "v"(v), "s"(base), "n"(q * 1024) : "memory");
            } else {
                if (PAD) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1"
                                      : : "v"(unit16 ^ (32u * q)), "v"(v), "s"(base), "n"(q * 1024) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3"
                                  : : "v"(unit16 ^ (32u * q)), "v"(v), "s"(base), "n"(q * 1024) : "memory");
            }
        }
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ReLU sign bits of a layer: dword fb>>1, bit 16*(fb&1)+r  <-  blk[fb][r] > 0.  blk is POST-ReLU (>= +0), so
// "> 0" is "bit pattern != 0": min(bits, 1) shifted into place is two vector instructions per value, where
// compare + select costs three plus the wait states gfx950 wants between a VCC write and its vector reader.
template <int NFB>
__device__ __forceinline__ void save_mask(float *mask_plane, int64_t m, int h, const f32x16 *blk) {
    u32x4 bits = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
        unsigned w = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            unsigned t;
            asm("v_min_u32 %0, 1, %1" : "=v"(t) : "v"(blk[fb][r]));
            w |= t << r;
        }
        bits[fb >> 1] |= w << (16 * (fb & 1));
    }
    reinterpret_cast<u32x4 *>(mask_plane)[2 * m + h] = bits;
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


}  // namespace mlp
