// Per-ray device code shared by the stand-alone sampling / integral kernels (sampling.hip, composite.hip) and
// the fused render pass (render_fused.hip): ONE wavefront works on ONE ray, its sample row lives in LDS.
// Everything that decides an integer in the reference (the pdf normaliser, the cdf, the bin search) follows
// ATen's CPU evaluation order literally, with separately rounded fp32 operations, so that bin indices are
// bit-identical to the reference's CPU path whichever kernel runs this code.
#pragma once
#include "common.h"

// X_FUSED_TIMELINE builds: RD_STAMP() records the cycle counter of thread 0 of workgroup 0 (render_fused.hip)
#ifndef RD_STAMP
#define RD_STAMP() do {} while (0)
#endif

namespace render {

// LDS traffic of one wavefront is processed in program order, so a row written by the wave is readable by any
// of its lanes afterwards; only the compiler has to be kept from reordering across the phase boundary.
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- cross-lane moves on the vector ALU (DPP / permlane swaps): a ds_bpermute-based __shfl costs an LDS round trip
// (~100+ cycles in a dependent chain: the 21 exchange stages of the sort and the scans of a ray were ~5 k cycles of it)
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_mov(unsigned old, unsigned src) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, BANK_MASK, true);
}
// value of lane (lane ^ D), D a power of two
template <int D>
__device__ __forceinline__ unsigned lane_xor(unsigned v, int lane) {
    if (D == 1) return dpp_mov<0xB1>(0u, v);                      // quad_perm [1,0,3,2]
    if (D == 2) return dpp_mov<0x4E>(0u, v);                      // quad_perm [2,3,0,1]
    if (D == 4) return dpp_mov<0x114, 0xF, 0xA>(dpp_mov<0x104, 0xF, 0x5>(0u, v), v);   // row_shl:4 into banks 0,2; row_shr:4 into 1,3
    if (D == 8) return dpp_mov<0x128>(0u, v);                     // row_ror:8
    if (D == 16) {   // v_permlane16_swap: odd rows of the first operand <-> even rows of the second
        const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? p[0] : p[1];
    }
    const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // upper half of the first <-> lower half of the second
    return (lane & 32) ? p[0] : p[1];
}

__device__ __forceinline__ int ceil_log2_i(int x) {
    if (x <= 2) return 1;
    return 32 - __builtin_clz((unsigned)(x - 1));
}

// ATen's CPU sum over a contiguous last dimension (SumKernel.cpp: vectorized_inner_sum ->
// row_sum -> multi_row_sum, 8-wide vectors, 4 interleaved accumulators, 4-level cascade).
// Lane l < 8 plays vector lane l and returns its partial; the caller finishes on lane 0.
__device__ inline float aten_sum_vector_lane(const float *row, int size0, int l) {
    constexpr int VEC = 8, ILP = 4, LEVELS = 4;
    const int vec_size = size0 / VEC;
    const int size_ilp = vec_size / ILP;
    int level_power = ceil_log2_i(size_ilp) / LEVELS;
    if (level_power < 4) level_power = 4;
    const int level_step = 1 << level_power;
    const int level_mask = level_step - 1;
    float acc[LEVELS][ILP];
#pragma unroll
    for (int j = 0; j < LEVELS; ++j)
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc[j][k] = 0.0f;
    int i = 0;
    for (; i + level_step <= size_ilp;) {
        for (int j = 0; j < level_step; ++j, ++i) {
#pragma unroll
            for (int k = 0; k < ILP; ++k)
                acc[0][k] = __fadd_rn(acc[0][k], row[(i * ILP + k) * VEC + l]);
        }
        bool stop = false;
#pragma unroll
        for (int j = 1; j < LEVELS; ++j) {
            if (!stop) {
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    acc[j][k] = __fadd_rn(acc[j][k], acc[j - 1][k]);
                    acc[j - 1][k] = 0.0f;
                }
                const int mask = level_mask << (j * level_power);
                if ((i & mask) != 0) stop = true;
            }
        }
    }
    for (; i < size_ilp; ++i) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc[0][k] = __fadd_rn(acc[0][k], row[(i * ILP + k) * VEC + l]);
    }
#pragma unroll
    for (int j = 1; j < LEVELS; ++j)
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc[0][k] = __fadd_rn(acc[0][k], acc[j][k]);
    // row_sum tail: whole vectors left over after the (-1, ILP) view
    for (int v = size_ilp * ILP; v < vec_size; ++v) acc[0][0] = __fadd_rn(acc[0][0], row[v * VEC + l]);
#pragma unroll
    for (int k = 1; k < ILP; ++k) acc[0][0] = __fadd_rn(acc[0][0], acc[0][k]);
    return acc[0][0];
}

__device__ __forceinline__ double wave_inclusive_scan(double v, int lane);

// LDS floats one ray needs besides its S sorted sample positions: unsorted positions, weights / pdf, cdf, partials
__host__ __device__ constexpr int hierarchical_scratch_floats(int Sc, int Sf) {
    return (((Sc + Sf) + 3) & ~3) + 2 * ((Sc + 3) & ~3) + 16;   // every row starts 16-byte aligned
}

// stratified_sampler.py:107-109 (coarse branch): t = t_bins + partition_size * U1, into the LDS row `t`
__device__ __forceinline__ void stratified_ray(int lane, int S, const float *__restrict__ t_bins, float ps,
                                               const float *__restrict__ u1_row, float *t) {
    for (int s = lane; s < S; s += WAVE) t[s] = __fadd_rn(t_bins[s], __fmul_rn(ps, u1_row[s]));
    wave_fence();
}

// ---- torch.sort of one row of S floats (ascending) as a rank sort: out[rank(e)] = in[e], rank(e) = number of
// elements that sort before e.  The sorted VALUES do not depend on how ties are ordered, but ranks must be
// distinct, so ties are broken by position: element j sorts before e iff (key(in[j]), j) < (key(in[e]), e) with
// key() the usual order-preserving map of float bits to unsigned -- ONE 64-bit unsigned compare per pair.  A lane
// owns K elements per sweep (e = e0 + lane + 64 k) and ranks them in one walk over the row, four positions per
// LDS read: for S = 192, 48 broadcast reads and 48 x 4 x 3 compares per lane.  (-0.0 sorts before +0.0 here,
// torch.sort leaves them in input order: the only difference, and not one in the VALUE sequence's use.)
__device__ __forceinline__ unsigned sort_key(float v) {
    const unsigned b = __float_as_uint(v);
    return b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
}

template <int K>
__device__ __forceinline__ void rank_sort_sweep(int lane, int S, int e0, const float *in, float *out) {
    float x[K];
    unsigned long long key[K];
    int rank[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = e0 + lane + WAVE * k;
        x[k] = e < S ? in[e] : 0.0f;
        key[k] = ((unsigned long long)sort_key(x[k]) << 32) | (unsigned)e;
        rank[k] = 0;
    }
    const int S4 = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) ? (S & ~3) : 0;
#pragma unroll 2
    for (int j = 0; j < S4; j += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(in + j);
        const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned long long kj = ((unsigned long long)sort_key(xs[c]) << 32) | (unsigned)(j + c);
#pragma unroll
            for (int k = 0; k < K; ++k) rank[k] += kj < key[k] ? 1 : 0;
        }
    }
    for (int j = S4; j < S; ++j) {
        const unsigned long long kj = ((unsigned long long)sort_key(in[j]) << 32) | (unsigned)j;
#pragma unroll
        for (int k = 0; k < K; ++k) rank[k] += kj < key[k] ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (e0 + lane + WAVE * k < S) out[rank[k]] = x[k];
}

// The common case has no ties: then rank(e) = #{j : in[j] < in[e]} -- one full-rate float compare per pair (the
// 64-bit compare of the exact sweep runs at a quarter of that).  Ties (or NaNs) make two elements claim one slot
// and leave another empty; the row is therefore pre-filled with NaN, and if any slot is still NaN after the
// scatter the whole row is redone by the exact sweep.
template <int K>
__device__ __forceinline__ bool rank_sort_sweep_fast(int lane, int S, int e0, const float *in, float *out) {
    float x[K];
    int rank[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = e0 + lane + WAVE * k;
        x[k] = e < S ? in[e] : 0.0f;
        rank[k] = 0;
    }
    const int S4 = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) ? (S & ~3) : 0;
#pragma unroll 2
    for (int j = 0; j < S4; j += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(in + j);
        const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < K; ++k) rank[k] += xs[c] < x[k] ? 1 : 0;
    }
    for (int j = S4; j < S; ++j) {
        const float xj = in[j];
#pragma unroll
        for (int k = 0; k < K; ++k) rank[k] += xj < x[k] ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (e0 + lane + WAVE * k < S) out[rank[k]] = x[k];
    return true;
}

// ---- S <= 256: bitonic network in registers, four keys per lane (element e = 4 lane + k).  The keys are the
// order-preserving unsigned images of the floats (sort_key), so min / max are single integer instructions and every
// bit pattern has its place: the sorted VALUE sequence is the one torch.sort returns (ties are equal values; the rank
// sort above orders -0.0 / +0.0 the same way).  Strides 1 and 2 stay inside a lane; a stride >= 4 exchanges with lane
// ^ (stride / 4).  36 compare-exchange stages, ~0.5 k instructions -- the O(S^2) rank sweep costs ~10 k cycles per ray.
// Rows with a NaN keep the rank sort (torch.sort puts every NaN last, whatever its sign bit).
__device__ __forceinline__ float sort_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
__device__ __forceinline__ void cmpx(unsigned &a, unsigned &b, bool ascending) {
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    a = ascending ? lo : hi;
    b = ascending ? hi : lo;
}
__device__ __forceinline__ bool bitonic_sort_row(int lane, int S, const float *in, float *out) {
    unsigned x[4];
    bool nan = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = 4 * lane + k;
        const float v = e < S ? in[e] : 0.0f;
        nan |= v != v;
        x[k] = e < S ? sort_key(v) : 0xFFFFFFFFu;       // padding sorts behind everything
    }
    if (__any(nan)) return false;
#pragma unroll
    for (int size = 2; size <= 256; size <<= 1) {
        // direction of the bitonic run this element sits in (the last merge, size 256, is ascending everywhere)
        const bool up_lane = size >= 256 ? true : ((4 * lane) & size) == 0;      // size >= 4: a property of the lane
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 4) {
                const int d = stride >> 2;
                const bool low = (lane & d) == 0;                                // this lane holds the lower element
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned y = stride == 4 ? lane_xor<1>(x[k], lane) : stride == 8 ? lane_xor<2>(x[k], lane)
                                     : stride == 16 ? lane_xor<4>(x[k], lane) : stride == 32 ? lane_xor<8>(x[k], lane)
                                     : stride == 64 ? lane_xor<16>(x[k], lane) : lane_xor<32>(x[k], lane);
                    const unsigned lo = x[k] < y ? x[k] : y, hi = x[k] < y ? y : x[k];
                    x[k] = (low == up_lane) ? lo : hi;
                }
            } else if (stride == 2) {
                cmpx(x[0], x[2], up_lane);
                cmpx(x[1], x[3], up_lane);
            } else {   // stride 1; size 2: the direction alternates with bit 1 of the element index
                cmpx(x[0], x[1], size == 2 ? true : up_lane);
                cmpx(x[2], x[3], size == 2 ? false : up_lane);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (4 * lane + k < S) out[4 * lane + k] = sort_unkey(x[k]);
    return true;
}

__device__ __forceinline__ void rank_sort_row(int lane, int S, const float *in, float *out) {
    if (S <= 4 * WAVE && bitonic_sort_row(lane, S, in, out)) return;
    if (S <= 4 * WAVE) {   // (a row with NaNs) one sweep covers the row: try the tie-free fast path first
        const float hole = __uint_as_float(0x7fc00000u);
        for (int e = lane; e < S; e += WAVE) out[e] = hole;
        wave_fence();
        if (S > 3 * WAVE) rank_sort_sweep_fast<4>(lane, S, 0, in, out);
        else if (S > 2 * WAVE) rank_sort_sweep_fast<3>(lane, S, 0, in, out);
        else if (S > WAVE) rank_sort_sweep_fast<2>(lane, S, 0, in, out);
        else rank_sort_sweep_fast<1>(lane, S, 0, in, out);
        wave_fence();
        bool bad = false;
        for (int e = lane; e < S; e += WAVE) bad |= out[e] != out[e];
        if (!__any(bad)) return;
        wave_fence();
    }
    for (int e0 = 0; e0 < S; e0 += 4 * WAVE) {
        const int left = S - e0;
        if (left > 3 * WAVE) rank_sort_sweep<4>(lane, S, e0, in, out);
        else if (left > 2 * WAVE) rank_sort_sweep<3>(lane, S, e0, in, out);
        else if (left > WAVE) rank_sort_sweep<2>(lane, S, e0, in, out);
        else rank_sort_sweep<1>(lane, S, e0, in, out);
    }
}

// stratified_sampler.py:57-90 + ray_samplers/utils.py:8-58 for one ray: floor the coarse weights in place,
// inverse-CDF pick of Sf bins, in-bin jitter, sort(cat[coarse, fine]) into the LDS row `t_srt` (Sc + Sf floats).
// `scratch`: hierarchical_scratch_floats(Sc, Sf) LDS floats private to this wavefront.
__device__ __forceinline__ void hierarchical_ray(int lane, int Sc, int Sf, const float *__restrict__ t_bins, float ps,
                                        float *__restrict__ weights_row, const float *__restrict__ u1_row,
                                        const float *__restrict__ u2_row, const float *__restrict__ u3_row,
                                        int64_t *__restrict__ bin_idx_row, float *scratch, float *t_srt) {
    const int S = Sc + Sf;
    // the fine draws of the first search round, requested before anything else: their HBM latency (~2 k cycles in
    // front of the search) runs under the pdf / cdf phases
    const bool pre_a = lane < Sf, pre_b = lane + WAVE < Sf;
    const float pre_ya = pre_a ? u2_row[lane] : 0.0f, pre_yb = pre_b ? u2_row[lane + WAVE] : 0.0f;
    const float pre_ja = pre_a ? u3_row[lane] : 0.0f, pre_jb = pre_b ? u3_row[lane + WAVE] : 0.0f;
    float *t_raw = scratch;                  // S   coarse then fine, unsorted
    float *w = t_raw + ((S + 3) & ~3);       // Sc  weights + 1e-5, then pdf
    float *cdf = w + ((Sc + 3) & ~3);        // Sc
    float *part = cdf + ((Sc + 3) & ~3);     // 8 partial sums + 1 normaliser
    // utils.py:31  weights += 1e-5 (in place, visible to the caller)
    for (int s = lane; s < Sc; s += WAVE) {
        const float v = __fadd_rn(weights_row[s], 1e-5f);
        weights_row[s] = v;
        w[s] = v;
        // stratified_sampler.py:77  new coarse jitter
        t_raw[s] = __fadd_rn(t_bins[s], __fmul_rn(ps, u1_row[s]));
    }
    wave_fence();
    RD_STAMP();
    // utils.py:32  normalizer = torch.sum(weights, -1) in ATen's order
    if (lane < 8) part[lane] = aten_sum_vector_lane(w, Sc, lane);
    wave_fence();
    if (lane == 0) {
        float fin = 0.0f;
        for (int k = (Sc / 8) * 8; k < Sc; ++k) fin = __fadd_rn(fin, w[k]);
        for (int l = 0; l < 8; ++l) fin = __fadd_rn(fin, part[l]);
        part[8] = fin;
    }
    wave_fence();
    const float norm = part[8];
    for (int s = lane; s < Sc; s += WAVE) w[s] = __fdiv_rn(w[s], norm);  // utils.py:33
    wave_fence();
    RD_STAMP();
    // utils.py:36-40  cdf = [0, cumsum(pdf)[:-1]] ; ATen CPU cumsum: ONE double accumulator walking the row,
    // every prefix rounded to fp32.  The chain of additions is sequential by definition (a parallel scan
    // rounds differently once the pdf spans more than ~2^29); what need not be sequential is the memory traffic:
    // every lane takes one pdf value into a register, the chain reads them with v_readlane (wave-uniform, so all
    // lanes run the same chain and lane l simply keeps prefix l), and the cdf row is written once at the end.
    {
        // When every pdf value is a positive normal number and the row's exponents span few enough bits, EVERY partial
        // sum of the row is exactly representable in a double (53 bits >= span + 24 + log2(Sc)): additions in any
        // order and grouping give the sequential chain's doubles, hence its fp32 roundings -- a shuffle scan then
        // replaces the 64-step dependent chain (4.1 k of the 23.5 k cycles four rays cost).  Weights of a rendered ray
        // lie in [1e-5, 1 + 1e-5]: 17 bits.  Anything else (zeros, denormals, negatives, NaN, a huge span) walks the chain.
        int emin = 255, emax = 0;
        bool plain = true;
        for (int s = lane; s < Sc; s += WAVE) {
            const unsigned b = __float_as_uint(w[s]);
            const int e = (int)((b >> 23) & 0xFFu);
            plain &= (b >> 31) == 0 && e != 0 && e != 255;
            emin = e < emin ? e : emin;
            emax = e > emax ? e : emax;
        }
        {   // butterfly over the wave: every lane ends with the row's extremes
            auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
            unsigned hi = (unsigned)emax, lo = (unsigned)(255 - emin);
            hi = mx(hi, lane_xor<1>(hi, lane)); lo = mx(lo, lane_xor<1>(lo, lane));
            hi = mx(hi, lane_xor<2>(hi, lane)); lo = mx(lo, lane_xor<2>(lo, lane));
            hi = mx(hi, lane_xor<4>(hi, lane)); lo = mx(lo, lane_xor<4>(lo, lane));
            hi = mx(hi, lane_xor<8>(hi, lane)); lo = mx(lo, lane_xor<8>(lo, lane));
            hi = mx(hi, lane_xor<16>(hi, lane)); lo = mx(lo, lane_xor<16>(lo, lane));
            hi = mx(hi, lane_xor<32>(hi, lane)); lo = mx(lo, lane_xor<32>(lo, lane));
            emax = (int)hi; emin = 255 - (int)lo;
        }
        const bool exact = !__any(!plain) && (emax - emin) + ceil_log2_i(Sc) <= 27;
        double run = 0.0;
        if (exact) {
            for (int base = 0; base < Sc; base += WAVE) {
                const double mine = (base + lane < Sc) ? (double)w[base + lane] : 0.0;
                const double incl = wave_inclusive_scan(mine, lane);
                if (base + lane < Sc) cdf[base + lane] = (float)(run + (incl - mine));   // sum of pdf[0 .. base + lane - 1]
                run += __shfl(incl, WAVE - 1, WAVE);
            }
        } else
        for (int base = 0; base < Sc; base += WAVE) {
            const float mine = (base + lane < Sc) ? w[base + lane] : 0.0f;
            float keep = 0.0f;   // cdf[base + lane] = sum of pdf[0 .. base + lane - 1]
            const int cnt = (Sc - base) < WAVE ? (Sc - base) : WAVE;
            if (cnt == WAVE) {
#pragma unroll
                for (int k = 0; k < WAVE; ++k) {
                    if (lane == k) keep = (float)run;
                    run += (double)__shfl(mine, k, WAVE);
                }
            } else {
                for (int k = 0; k < cnt; ++k) {
                    if (lane == k) keep = (float)run;
                    run += (double)__shfl(mine, k, WAVE);
                }
            }
            if (base + lane < Sc) cdf[base + lane] = keep;
        }
    }
    wave_fence();
    RD_STAMP();
    // utils.py:43-56  searchsorted(right=True) - 1, gather, in-bin jitter.  searchsorted(right=True) = the number of
    // cdf entries <= y; the cdf is non-decreasing (prefix sums of a positive pdf), so the count is found by
    // descending power-of-two steps -- branch-free, two fine samples per lane side by side (their LDS reads
    // overlap), the draws requested before the first probe.
    {
        int top = 1;
        while (2 * top <= Sc) top *= 2;
        for (int f0 = 0; f0 < Sf; f0 += 2 * WAVE) {
            const int fa = f0 + lane, fb = fa + WAVE;
            const bool has_a = fa < Sf, has_b = fb < Sf;
            const float ya = f0 == 0 ? pre_ya : (has_a ? u2_row[fa] : 0.0f), yb = f0 == 0 ? pre_yb : (has_b ? u2_row[fb] : 0.0f);
            const float ja = f0 == 0 ? pre_ja : (has_a ? u3_row[fa] : 0.0f), jb = f0 == 0 ? pre_jb : (has_b ? u3_row[fb] : 0.0f);
            int pa = 0, pb = 0;   // entries known to be <= y
            for (int step = top; step > 0; step >>= 1) {
                const int na = pa + step, nb = pb + step;
                const float ca = cdf[(na <= Sc ? na : Sc) - 1], cb2 = cdf[(nb <= Sc ? nb : Sc) - 1];
                if (na <= Sc && ca <= ya) pa = na;
                if (nb <= Sc && cb2 <= yb) pb = nb;
            }
            if (has_a) {
                int k = pa - 1;
                if (bin_idx_row) bin_idx_row[fa] = (int64_t)k;
                if (k < 0) k = 0;
                t_raw[Sc + fa] = __fadd_rn(t_bins[k], __fmul_rn(ps, ja));
            }
            if (has_b) {
                int k = pb - 1;
                if (bin_idx_row) bin_idx_row[fb] = (int64_t)k;
                if (k < 0) k = 0;
                t_raw[Sc + fb] = __fadd_rn(t_bins[k], __fmul_rn(ps, jb));
            }
        }
    }
    wave_fence();
    RD_STAMP();
    // stratified_sampler.py:87-90  sort(cat[coarse, fine])
    rank_sort_row(lane, S, t_raw, t_srt);
    wave_fence();
    RD_STAMP();
}

// ---- quadrature_integrator.py:41-65 for one ray: lane = sample, 64 samples per step.  The exclusive prefix sum of
// sigma * delta is a wave-level scan with lane shuffles in double (ATen's CPU cumsum accumulates in double),
// carried across steps.
// inclusive prefix sum over the 64 lanes: the DPP scan (row_shr 1, 2, 3; row_shr:4 / :8 into the upper banks;
// row_bcast 15 / 31 across rows), the two halves of a double moved separately
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = dpp_mov<CTRL, ROW_MASK, BANK_MASK>(0u, (unsigned)b);
    const unsigned hi = dpp_mov<CTRL, ROW_MASK, BANK_MASK>(0u, (unsigned)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_inclusive_scan(double v, int lane) {
    (void)lane;
    double s = v + dpp_mov_f64<0x111, 0xF, 0xF>(v);      // + lane - 1   (zeros shift into a row)
    s += dpp_mov_f64<0x112, 0xF, 0xF>(v);                // + lane - 2
    s += dpp_mov_f64<0x113, 0xF, 0xF>(v);                // + lane - 3
    s += dpp_mov_f64<0x114, 0xF, 0xE>(s);                // groups of 4 -> 8 (banks 1..3)
    s += dpp_mov_f64<0x118, 0xF, 0xC>(s);                // -> 16 (banks 2, 3)
    s += dpp_mov_f64<0x142, 0xA, 0xF>(s);                // row_bcast:15 into rows 1, 3
    s += dpp_mov_f64<0x143, 0xC, 0xF>(s);                // row_bcast:31 into rows 2, 3
    return s;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// sigma(s), delta(s), radiance(s, c) are accessors (global rows in composite.hip, LDS rows in the fused pass);
// weights_row is written for s < S; the three colour sums come back in all lanes.
template <class Sigma, class Delta, class Radiance>
__device__ __forceinline__ void composite_ray(int lane, int S, Sigma sigma, Delta delta, Radiance radiance,
                                              float *__restrict__ weights_row, float (&rgb)[3]) {
    double carry = 0.0;  // sum of tau over all earlier 64-sample steps
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
    for (int s0 = 0; s0 < S; s0 += WAVE) {
        const int s = s0 + lane;
        const bool live = s < S;
        const float tau = live ? sigma(s) * delta(s) : 0.0f;  // quadrature_integrator.py:41
        const double incl = wave_inclusive_scan((double)tau, lane);
        double excl = __shfl_up(incl, 1, WAVE);
        if (lane == 0) excl = 0.0;
        // :44-52  T_i = exp(-cumsum([0, tau])[:-1]) ; the prefix is rounded to fp32 like ATen's
        const float T = expf(-(float)(carry + excl));
        const float alpha = 1.0f - expf(-tau);  // :55
        const float w = T * alpha;              // :58
        if (live) {
            weights_row[s] = w;
            acc0 += w * radiance(s, 0);  // :62-65
            acc1 += w * radiance(s, 1);
            acc2 += w * radiance(s, 2);
        }
        carry += __shfl(incl, WAVE - 1, WAVE);
    }
    rgb[0] = wave_sum(acc0);
    rgb[1] = wave_sum(acc1);
    rgb[2] = wave_sum(acc2);
}

}  // namespace render
