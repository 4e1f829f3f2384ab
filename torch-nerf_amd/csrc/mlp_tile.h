// One 128-sample tile of the fused encode + NeRF forward (a8 + a9 + a10), shared by the plain MLP kernel
// (mlp_forward.hip) and the fused render pass (render_fused.hip).  See mlp_forward.hip for the structure.
#pragma once
#include "mlp_device.h"

namespace mlp {

// X_TIMELINE builds (scripts/timeline.py): cycle stamps of one lane, written to a buffer the kernel provides
struct Timeline {
#ifdef X_TIMELINE
    unsigned long long *buf;
    int n;
    bool on;
    __device__ __forceinline__ void stamp() { if (on) buf[n++] = __builtin_readcyclecounter(); }
#else
    __device__ __forceinline__ void stamp() {}
#endif
};
#define NERF_TS() tl.stamp()

// How the tile obtains its two encodings:
//   IN_ENCODED  pos (M, e_pos), view_dir (M, e_dir) rows already encoded (plain NeRF.forward, any encoder)
//   IN_SHIPPED  raw points; PositionalEncoder(3, 10, True) / (3, 4, True) -- the reference's shipped yaml -- from the
//               compile-time table (one sincos per octave and channel, every index a constant)
//   IN_LEVELS   raw points; PositionalEncoder(3, net.l_pos, net.inc_pos) / (3, net.l_dir, net.inc_dir) with run-time
//               levels: enc_feature per fragment register (48 sincos + index arithmetic per lane and tile instead of
//               39), same sincos_cw, so the shipped configuration gives the same bits either way
enum { IN_ENCODED = 0, IN_SHIPPED = 1, IN_LEVELS = 2 };

// Inputs of this lane's sample: raw[0..2] position, raw[3..5] direction (IN_ENCODED: read from pos/dir rows mc instead).
// `after_encode()` runs once the raw inputs are consumed (the caller prefetches the next tile's there).
// Outputs: sigma and the three colours of sample m, valid in the lanes of both halves.
// RAYDIR (render_fused.hip): the direction of this lane's sample is its RAY's, so fc_9's direction contribution was
// computed once per ray: `fc9_init` points at this lane's ray row of 128 floats (LDS) = fc_9.bias + W9[:, 256:] enc(d),
// accumulated in the k order of the MFMA chain below -- the accumulators start from it, no direction is encoded, and
// the direction pair of the stream is neither fetched (Pipe::skip_mask) nor multiplied.  Same bits, 64 MFMAs, one
// acquire and twelve sincos per sample less.
template <int INPUT, bool SAVE, bool RAYDIR = false, class AfterEncode>
__device__ __forceinline__ void forward_tile(const Net &net, const float (&raw)[6], const float *__restrict__ pos,
                                             const float *__restrict__ dir, int64_t mc, int64_t m, int64_t MP,
                                             int h, Pipe &pipe, const char *lds, const float *cb, const int (&offq)[4],
                                             float *__restrict__ saved, AfterEncode after_encode, Timeline &tl,
                                             float &sigma_result, float (&y)[3], const float *fc9_init = nullptr) {
    static_assert(!RAYDIR || (!SAVE && INPUT != IN_ENCODED), "ray-constant directions: raw-input inference only");
    constexpr bool ENCODED = INPUT == IN_ENCODED;
    const int E_POS = INPUT == IN_SHIPPED ? DEFAULT_NET.e_pos : net.e_pos;
    const int E_DIR = INPUT == IN_SHIPPED ? DEFAULT_NET.e_dir : net.e_dir;
    const int L_POS = INPUT == IN_SHIPPED ? DEFAULT_NET.l_pos : net.l_pos;
    const int L_DIR = INPUT == IN_SHIPPED ? DEFAULT_NET.l_dir : net.l_dir;
    // ---- encodings, straight into B-fragment layout: reg r <-> feature 32 kb + (r&3) + 8 (r>>2) + 4 h
    f32x16 pe[2], de;
    if (ENCODED) {
        const float *prow = pos + mc * E_POS, *drow = dir + mc * E_DIR;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
            pe[0][r] = (k < E_POS) ? prow[k] : 0.0f;
            pe[1][r] = (32 + k < E_POS) ? prow[32 + k] : 0.0f;
            de[r] = (k < E_DIR) ? drow[k] : 0.0f;
        }
    } else if (INPUT == IN_LEVELS) {
        // run-time levels; a wave with a huge (or non-finite) coordinate takes the library sin/cos (wave-uniform choice)
        if (__builtin_expect(__any(encoding_needs_exact(raw, L_POS, L_DIR)), 0)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
                pe[0][r] = enc_feature<true>(k, raw[0], raw[1], raw[2], E_POS, net.inc_pos);
                pe[1][r] = enc_feature<true>(32 + k, raw[0], raw[1], raw[2], E_POS, net.inc_pos);
                if (!RAYDIR) de[r] = enc_feature<true>(k, raw[3], raw[4], raw[5], E_DIR, net.inc_dir);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
                pe[0][r] = enc_feature<false>(k, raw[0], raw[1], raw[2], E_POS, net.inc_pos);
                pe[1][r] = enc_feature<false>(32 + k, raw[0], raw[1], raw[2], E_POS, net.inc_pos);
                if (!RAYDIR) de[r] = enc_feature<false>(k, raw[3], raw[4], raw[5], E_DIR, net.inc_dir);
            }
        }
    } else if (__builtin_expect(__any(encoding_needs_exact(raw, L_POS, L_DIR)), 0)) {
        // some lane of this wave has a huge (or non-finite) coordinate: library sin/cos for the tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
            pe[0][r] = enc_feature<true>(k, raw[0], raw[1], raw[2], E_POS);
            pe[1][r] = enc_feature<true>(32 + k, raw[0], raw[1], raw[2], E_POS);
            if (!RAYDIR) de[r] = enc_feature<true>(k, raw[3], raw[4], raw[5], E_DIR);
        }
    } else {
        {
            float F[64];
            encode_table<DEFAULT_NET.l_pos, 64>(raw[0], raw[1], raw[2], F);
            table_to_fragment(F, 0, h, pe[0]);
            table_to_fragment(F, 1, h, pe[1]);
        }
        if (!RAYDIR) {
            float F[32];
            encode_table<DEFAULT_NET.l_dir, 32>(raw[3], raw[4], raw[5], F);
            table_to_fragment(F, 0, h, de);
        }
    }
    after_encode();

    f32x16 acc[8], act[8];

    // ---- fc_in (nerf.py:102): one pair = the two 32-wide halves of the encoded position
    NERF_TS();
    {
        const char *w = lds + pipe.acquire();
        if (SAVE) {
            save_plane<2>(saved + pl_pe(MP), 64, m, h, pe);
            save_plane<1>(saved + pl_de(MP), 32, m, h, &de);
        }
        load_bias<8>(acc, cb + CB_BIAS, h);
        mma_pair<8>(acc, pe[0], pe[1], w, offq, pipe);
    }

    // ---- fc_1 .. fc_8 (nerf.py:103-113); skip connection at fc_5 (pos FIRST, :108).
    // The ReLU / record / bias section of layer l-1 sits AFTER the acquire of layer l's first
    // pair, so its vector-ALU work and stores overlap the DMA wait and the first MFMAs.
    float sigma_pre = 0.0f;
    for (int l = 1; l <= 8; ++l) {
        NERF_TS();
        const char *w = lds + pipe.acquire();
        NERF_TS();
#pragma unroll
        for (int fb = 0; fb < 8; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) act[fb][r] = relu1(acc[fb][r]);  // ReLU of layer l-1
        // record: h(l-1) leaves in 32 one-KiB stores spread over the first chunks of this layer's four pairs (its blocks
        // are the B operands of exactly these pairs: untouched until the next seam) -- at the seam they cost ~190 cycles each
        PlaneStore st;
        if (SAVE) {
            st.open(saved + pl_h(MP, l - 1), 256, m, h, act);
            save_mask<8>(saved + pl_masks(MP) + (int64_t)(l - 1) * MP * 8, m, h, act);
        }
        if (l == 8) sigma_pre = half_dot<8>(cb + CB_W8ROW0, act, h);  // density row of fc_8
        load_bias<8>(acc, l < 8 ? cb + CB_BIAS + l * 256 : cb + CB_BIAS8, h);
        NERF_TS();
        if (l == 5) {
            mma_pair<8>(acc, pe[0], pe[1], w, offq, pipe);
            w = lds + pipe.acquire();
        }
        mma_pair<8, false, SAVE ? 8 : 0>(acc, act[0], act[1], w, offq, pipe, &st, 0);
        NERF_TS();
#pragma unroll
        for (int pr = 1; pr < 4; ++pr) {
            w = lds + pipe.acquire();
            mma_pair<8, false, SAVE ? 8 : 0>(acc, act[2 * pr], act[2 * pr + 1], w, offq, pipe, &st, 8 * pr);
        }
    }

    // ---- fc_9 on cat([x[:,1:], view_dir]) (:116-118); fc_8 has no ReLU (:113).  The reference's cat puts the features
    // first; the ORDER OF ACCUMULATION here is bias, direction, features (any order is the same sum up to fp32
    // rounding; this one lets a ray-constant direction be folded into the starting value, see RAYDIR above)
    NERF_TS();
    {
        const char *w = lds + pipe.acquire();
#pragma unroll
        for (int fb = 0; fb < 8; ++fb) act[fb] = acc[fb];
        PlaneStore st;
        if (SAVE) st.open(saved + pl_y8(MP), 256, m, h, act);     // y8: spread over fc_9's four pairs
        if (RAYDIR) {
            load_bias<4>(acc, fc9_init, h);           // this lane's ray: bias + direction part, as the chain below leaves it
        } else {
            load_bias<4>(acc, cb + CB_BIAS9, h);
            mma_chunk<4, 0, 16>(acc, de, w, offq, &pipe);   // direction chunk + filler chunk (not multiplied)
            pipe.issue_done();
            w = lds + pipe.acquire();
        }
        mma_pair<4, false, SAVE ? 8 : 0>(acc, act[0], act[1], w, offq, pipe, &st, 0);
#pragma unroll
        for (int pr = 1; pr < 4; ++pr) {
            w = lds + pipe.acquire();
            mma_pair<4, false, SAVE ? 8 : 0>(acc, act[2 * pr], act[2 * pr + 1], w, offq, pipe, &st, 8 * pr);
        }
    }
    NERF_TS();
    sigma_pre += __shfl_xor(sigma_pre, 32, WAVE);
    sigma_result = fmaxf(sigma_pre + cb[CB_SCALARS], 0.0f);  // relu(x[:,0]) (:115)
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fb][r] = relu1(acc[fb][r]);
    if (SAVE) {
        save_plane<4>(saved + pl_h9(MP), 128, m, h, acc);
        save_mask<4>(saved + pl_masks(MP) + (int64_t)8 * MP * 8, m, h, acc);
    }

    // ---- fc_out + sigmoid (:119) on the vector ALU: 3 x 128 MACs per sample
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float p = half_dot<4>(cb + CB_WOUT + c * HALF, acc, h);
        p += __shfl_xor(p, 32, WAVE);
        y[c] = 1.0f / (1.0f + expf(-(p + cb[CB_SCALARS + 1 + c])));
    }
}

}  // namespace mlp
