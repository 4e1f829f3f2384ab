// Re-tiles the flat NeRF parameter blob (state_dict order) into the stream the fused
// MLP kernels DMA into LDS: const block + 78 forward chunks + 74 transposed chunks
// (layout: mlp_layout.h).  2.4 MB in, 4.8 MB out; runs once per parameter update.
#include "common.h"
#include "mlp_layout.h"
#include "net.h"

namespace {

using namespace mlp;

// (the per-element offset arithmetic is scalar-uniform in `layer` and runs once per parameter update: not hot)
struct Params {
    const float *P;
    Net net;
    int w_off[NUM_LAYERS], b_off[NUM_LAYERS], in[NUM_LAYERS];   // filled on the host (set)
    void set(const float *params) {
        P = params;
        for (int l = 0; l < NUM_LAYERS; ++l) {
            w_off[l] = (int)net.w_offset(l); b_off[l] = (int)net.b_offset(l); in[l] = net.layer_in(l);
        }
    }
    __device__ __forceinline__ float w(int layer, int n, int k) const { return P[w_off[layer] + n * in[layer] + k]; }
    __device__ __forceinline__ float b(int layer, int n) const { return P[b_off[layer] + n]; }
};

__device__ float forward_chunk_value(const Params &P, int ci, int n, int kk) {
    const int E_POS = P.net.e_pos, E_DIR = P.net.e_dir;
    if (ci < CH_TRUNK1) {
        const int k = 32 * ci + kk;
        return k < E_POS ? P.w(0, n, k) : 0.0f;
    }
    if (ci < CH_FC5_ENC) {
        const int l = 1 + (ci - CH_TRUNK1) / 8, kb = (ci - CH_TRUNK1) % 8;
        return P.w(l, n, 32 * kb + kk);
    }
    if (ci < CH_FC5) {
        const int k = 32 * (ci - CH_FC5_ENC) + kk;
        return k < E_POS ? P.w(5, n, k) : 0.0f;
    }
    if (ci < CH_TRUNK6) return P.w(5, n, E_POS + 32 * (ci - CH_FC5) + kk);
    if (ci < CH_FC8) {
        const int l = 6 + (ci - CH_TRUNK6) / 8, kb = (ci - CH_TRUNK6) % 8;
        return P.w(l, n, 32 * kb + kk);
    }
    if (ci < CH_FC9_DIR) return P.w(8, n + 1, 32 * (ci - CH_FC8) + kk);
    if (n >= HALF) return 0.0f;
    if (ci == CH_FC9_DIR) return kk < E_DIR ? P.w(9, n, FEAT + kk) : 0.0f;
    if (ci >= CH_FC9) return P.w(9, n, 32 * (ci - CH_FC9) + kk);
    return 0.0f;  // filler chunk (pairs)
}

// chunk of W^T: image row m = INPUT feature, k-group = 32 consecutive OUTPUT features
__device__ float backward_chunk_value(const Params &P, int ci, int m, int kk) {
    const int E_POS = P.net.e_pos, E_DIR = P.net.e_dir;
    if (ci < BW_FC9T) {   // slot-major: slot kb = direction block x output block kb
        const int slot = 8 * (ci - BW_DIRT) + (m >> 5), i = m & 31;
        return (slot < 4 && i < E_DIR) ? P.w(9, 32 * slot + kk, FEAT + i) : 0.0f;
    }
    if (ci < BW_FC8T) return P.w(9, 32 * (ci - BW_FC9T) + kk, m);
    if (ci < BW_FC7T) return P.w(8, 1 + 32 * (ci - BW_FC8T) + kk, m);
    if (ci < BW_FC5T) {
        const int l = 7 - (ci - BW_FC7T) / 8, cb = (ci - BW_FC7T) % 8;
        return P.w(l, 32 * cb + kk, m);
    }
    if (ci < BW_FC4T) return P.w(5, 32 * (ci - BW_FC5T) + kk, E_POS + m);
    if (ci < BW_FCINT) {
        const int l = 4 - (ci - BW_FC4T) / 8, cb = (ci - BW_FC4T) % 8;
        return P.w(l, 32 * cb + kk, m);
    }
    // slot-major: slot 2 kb + fb of fc_in^T / fc_5[:, :E_p]^T
    const int base = ci < BW_FC5POST ? BW_FCINT : BW_FC5POST;
    const int slot = 8 * (ci - base) + (m >> 5), k = 32 * (slot & 1) + (m & 31);
    return k < E_POS ? P.w(ci < BW_FC5POST ? 0 : 5, 32 * (slot >> 1) + kk, k) : 0.0f;
}

__device__ float const_block_value(const Params &P, int e) {
    if (e < CB_BIAS8) return P.b(e / 256, e % 256);
    if (e < CB_BIAS9) return P.b(8, 1 + (e - CB_BIAS8));
    if (e < CB_W8ROW0) return P.b(9, e - CB_BIAS9);
    if (e < CB_WOUT) return P.w(8, 0, e - CB_W8ROW0);
    if (e < CB_SCALARS) return P.w(10, (e - CB_WOUT) / HALF, (e - CB_WOUT) % HALF);
    if (e == CB_SCALARS) return P.b(8, 0);
    if (e < CB_SCALARS + 4) return P.b(10, e - CB_SCALARS - 1);
    return 0.0f;
}

__global__ void pack_kernel(const Params P, float *__restrict__ out) {
    const int64_t total = PACKED_BYTES / 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (e < CONST_FLOATS) {
            v = const_block_value(P, (int)e);
        } else {
            const int64_t r = e - CONST_FLOATS;
            const int ci = (int)(r / CHUNK_FLOATS);
            const int b = (int)(r % CHUNK_FLOATS) * 4;  // byte offset inside the chunk image
            const int n = b >> 7;                       // row (128 B per row)
            const int p = (b & 127) >> 4;               // physical 16-B slot
            const int c = p ^ ((n >> 1) & 7);           // logical k-group
            const int kk = 4 * c + ((b & 15) >> 2);
            v = ci < FWD_CHUNKS ? forward_chunk_value(P, ci, n, kk)
                                : backward_chunk_value(P, ci - FWD_CHUNKS, n, kk);
        }
        out[e] = v;
    }
}

// ---- bf16 stream (mlp_layout.h "bf16 inference stream"): value of row n, k-value kk (0..31) of chunk c of sub-step `sub`
__device__ float bf16_stream_value(const Params &P, int sub, int c, int n, int kk) {
    const int E_POS = P.net.e_pos;
    if (sub == 0 || sub == 17) {                       // encoded position into fc_in / fc_5 (pos first, nerf.py:108)
        const int k = 32 * c + kk;
        return k < E_POS ? P.w(sub == 0 ? 0 : 5, n, k) : 0.0f;
    }
    if (sub <= 16) return P.w(1 + (sub - 1) / 4, n, 32 * (2 * ((sub - 1) % 4) + c) + kk);
    if (sub <= 21) return P.w(5, n, E_POS + 32 * (2 * (sub - 18) + c) + kk);
    if (sub <= 29) return P.w(6 + (sub - 22) / 4, n, 32 * (2 * ((sub - 22) % 4) + c) + kk);
    return P.w(8, n + 1, 32 * (2 * (sub - 30) + c) + kk);   // subs 30..33: fc_8 rows 1..256
}
// fc_9 (subs 34..36): 128-row chunks, three per sub-step; chunk index ck = 3 (sub - 34) + c is the k-block, 8 = direction
__device__ float bf16_fc9_value(const Params &P, int ck, int n, int kk) {
    const int E_DIR = P.net.e_dir;
    if (ck < 8) return P.w(9, n, 32 * ck + kk);
    return (ck == 8 && kk < E_DIR) ? P.w(9, n, FEAT + kk) : 0.0f;
}

__global__ void pack_bf16_kernel(const Params P, char *__restrict__ out) {
    float *cblock = reinterpret_cast<float *>(out);
    __bf16 *stream = reinterpret_cast<__bf16 *>(out + CONST_BYTES);
    const int64_t n_bf16 = (int64_t)B16_SUBS * B16_SUB_BYTES / 2;
    const int64_t total = CONST_FLOATS + n_bf16;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        if (e < CONST_FLOATS) {
            cblock[e] = const_block_value(P, (int)e);
            continue;
        }
        const int64_t r = e - CONST_FLOATS;            // bf16 element index in the stream
        const int sub = (int)(r / (B16_SUB_BYTES / 2));
        const int in_sub = (int)(r % (B16_SUB_BYTES / 2)) * 2;   // byte offset inside the sub-step
        const int chunk_bytes = sub >= 34 ? B16_HALF_CHUNK_BYTES : B16_CHUNK_BYTES;
        const int c = in_sub / chunk_bytes;
        const int b = in_sub % chunk_bytes;                   // byte offset inside the chunk image
        const int n = b >> 6;                                 // row (64 B per row)
        const int slot = ((b & 63) >> 4) ^ ((n >> 2) & 3);    // logical fragment 2s+h
        const int el = (b & 15) >> 1;
        const int kk = 16 * (slot >> 1) + 8 * (el >> 2) + 4 * (slot & 1) + (el & 3);
        stream[r] = (__bf16)(sub >= 34 ? (c < 3 ? bf16_fc9_value(P, 3 * (sub - 34) + c, n, kk) : 0.0f)
                                       : bf16_stream_value(P, sub, c, n, kk));
    }
}


// ---- split-f16 stream (mlp_layout.h "split-f16 inference stream")
// The per-layer scale 2^s, s = 13 - floor(log2(max |W_l|)) (fc_8: rows 1..256 -- the density row stays fp32 on the vector
// ALU), in two steps because the stream is re-packed on EVERY call of the module (NeRF._stream_f16x2) and one block per
// layer walking 81 k weights cost 16 us of a 2.9 ms render step: F2_SLICES blocks per layer leave partial maxima in the
// const block's spare floats, and every block of the pack kernel folds the 160 partials into its own copy of the scales.
constexpr int F2_SLICES = 16;
constexpr int F2_CB_PARTIAL = F2_CB_SCALE + 10;     // 10 x F2_SLICES floats
static_assert(F2_CB_PARTIAL + 10 * F2_SLICES <= F2_CB_PLANE_MAX, "const block");

__global__ __launch_bounds__(256) void f16x2_absmax_kernel(const Params P, float *__restrict__ cblock) {
    const int l = blockIdx.y, slice = blockIdx.x;
    // the layer's weights are one contiguous run of the blob (fc_8: behind its first row)
    const float *w = P.P + P.w_off[l] + (l == 8 ? P.in[8] : 0);
    const int count = (l == 8 ? FEAT : Net::layer_out(l)) * P.in[l];
    const int per = (count + F2_SLICES - 1) / F2_SLICES, lo = slice * per, hi = lo + per < count ? lo + per : count;
    __shared__ float red[256];
    float m = 0.0f;
    for (int e = lo + threadIdx.x; e < hi; e += 256) m = fmaxf(m, fabsf(w[e]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) cblock[F2_CB_PARTIAL + l * F2_SLICES + slice] = red[0];
}

// scale[l] = 2^s, unscale[l] = 2^-s from the partial maxima (every thread of the block calls this; shared arrays of 10)
__device__ __forceinline__ void f16x2_scales(const float *cblock, float *scale, float *unscale) {
    if (threadIdx.x < 10) {
        float mx = 0.0f;
        for (int k = 0; k < F2_SLICES; ++k) mx = fmaxf(mx, cblock[F2_CB_PARTIAL + threadIdx.x * F2_SLICES + k]);
        int s = 0;
        if (mx > 0.0f && mx < INFINITY) {
            s = 13 - ilogbf(mx);
            s = s > 40 ? 40 : (s < -40 ? -40 : s);     // (all-tiny or huge layers: keep 2^s and the scaled biases finite)
        }
        scale[threadIdx.x] = ldexpf(1.0f, s);
        unscale[threadIdx.x] = ldexpf(1.0f, -s);
    }
    __syncthreads();
}

// weight value of stream position (sub, kbi = k-block inside the sub-step, row n, kk) and its layer; zero filler where the
// (padded) k-block has no input feature
__device__ float f16x2_stream_value(const Params &P, const F2Layout &L, int sub, int kbi, int n, int kk, int &layer) {
    const int E_POS = P.net.e_pos, E_DIR = P.net.e_dir;
    if (sub < L.sub_fc1() || (sub >= L.sub_fc5_pos() && sub < L.sub_fc5())) {   // encoded position into fc_in / fc_5 (pos first)
        layer = sub < L.sub_fc1() ? 0 : 5;
        const int k = 32 * (sub < L.sub_fc1() ? sub : sub - L.sub_fc5_pos()) + kk;
        return k < E_POS ? P.w(layer, n, k) : 0.0f;
    }
    if (sub < L.sub_fc5_pos()) { layer = 1 + (sub - L.sub_fc1()) / 8; return P.w(layer, n, 32 * ((sub - L.sub_fc1()) % 8) + kk); }
    if (sub < L.sub_fc6()) { layer = 5; return P.w(5, n, E_POS + 32 * (sub - L.sub_fc5()) + kk); }
    if (sub < L.sub_fc8()) { layer = 6 + (sub - L.sub_fc6()) / 8; return P.w(layer, n, 32 * ((sub - L.sub_fc6()) % 8) + kk); }
    if (sub < L.sub_fc9()) { layer = 8; return P.w(8, n + 1, 32 * (sub - L.sub_fc8()) + kk); }
    layer = 9;
    const int ck = 2 * (sub - L.sub_fc9()) + kbi;                     // k-block of fc_9; 8 .. 8 + NDIR - 1 = direction, then filler
    if (ck < 8) return P.w(9, n, 32 * ck + kk);
    const int k = 32 * (ck - 8) + kk;
    return k < E_DIR ? P.w(9, n, FEAT + k) : 0.0f;
}

// transposed stream (reverse chain): sub-step t of F2Layout::bwd_subs(), image row n = input feature, k-value kk of its k-block
__device__ float f16x2_bwd_stream_value(const Params &P, int t, int n, int kk, int &layer) {
    if (t < 4) { layer = 9; return P.w(9, 32 * t + kk, n); }                       // fc_9[:, 0:256]^T: 128 outputs
    layer = 8 - (t - 4) / 8;
    const int o = 32 * ((t - 4) % 8) + kk;                                         // output feature of the layer
    if (layer == 8) return P.w(8, 1 + o, n);                                       // rows 1..256 (row 0: density, vector ALU)
    if (layer == 5) return P.w(5, o, P.net.e_pos + n);                             // behind the skip connection's position block
    return P.w(layer, o, n);
}

__global__ void pack_f16x2_kernel(const Params P, char *__restrict__ out) {
    float *cblock = reinterpret_cast<float *>(out);
    __shared__ float scale[10], unscale[10];
    f16x2_scales(cblock, scale, unscale);
    _Float16 *stream = reinterpret_cast<_Float16 *>(out + CONST_BYTES);
    const F2Layout L = f2_layout(P.net.e_pos, P.net.e_dir);
    const int64_t n_f16 = (int64_t)(L.subs() + L.bwd_subs()) * F2_SUB_BYTES / 2;
    const int64_t total = CONST_FLOATS + n_f16;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        if (e < CONST_FLOATS) {
            if (e >= F2_CB_PARTIAL) continue;                                  // the absmax kernel's partials (and padding)
            float v;
            if (e >= F2_CB_SCALE) v = scale[e - F2_CB_SCALE];
            else if (e >= F2_CB_UNSCALE) v = unscale[e - F2_CB_UNSCALE];
            else {
                v = const_block_value(P, (int)e);
                if (e < CB_BIAS8) v *= scale[e / 256];                         // biases ride in the scaled accumulators
                else if (e < CB_BIAS9) v *= scale[8];
                else if (e < CB_W8ROW0) v *= scale[9];
            }
            cblock[e] = v;
            continue;
        }
        const int64_t r = e - CONST_FLOATS;                    // f16 element index in the stream
        const int sub = (int)(r / (F2_SUB_BYTES / 2));
        const int in_sub = (int)(r % (F2_SUB_BYTES / 2)) * 2;  // byte offset inside the sub-step
        const bool reverse = sub >= L.subs();
        const int image_bytes = (sub >= L.sub_fc9() && !reverse) ? F2_IMAGE_BYTES / 2 : F2_IMAGE_BYTES;
        const int image = in_sub / image_bytes;                // 2 kbi + part (0 hi, 1 lo)
        const int b = in_sub % image_bytes;                    // byte offset inside the image
        const int n = b >> 6;                                  // row (64 B per row)
        const int g = ((b & 63) >> 4) ^ f2_sigma((n >> 2) & 3);   // logical fragment = lane group (f2_frag_offset)
        const int el = (b & 15) >> 1;
        const int kk = 16 * (el >> 2) + 4 * g + (el & 3);
        int layer;
        const float w = (reverse ? f16x2_bwd_stream_value(P, sub - L.subs(), n, kk, layer)
                                 : f16x2_stream_value(P, L, sub, image >> 1, n, kk, layer)) * scale[layer];
        const _Float16 hi = (_Float16)w;
        stream[r] = (image & 1) ? (_Float16)(w - (float)hi) : hi;
    }
}

}  // namespace

NERF_API int nerf_mlp_path(const nerf_net_t *net) {
    nerf_net_t d;
    return nerf::net_describe(net, d);
}

NERF_API int64_t nerf_mlp_param_count(const nerf_net_t *net) {
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0) return -1;
    const int64_t F = d.feat_dim, P = d.pos_dim, V = d.view_dir_dim, H = F / 2;   // nerf.py:49-59
    return (P * F + F) + 4 * (F * F + F) + ((F + P) * F + F) + 2 * (F * F + F) + (F * (F + 1) + F + 1) +
           ((F + V) * H + H) + (H * 3 + 3);
}

NERF_API int64_t nerf_mlp_packed_bf16_bytes(const nerf_net_t *net) {
    mlp::Net n;
    return nerf::fused_net(net, n, "nerf_mlp_packed_bf16_bytes") == NERF_OK ? mlp::B16_PACKED_BYTES : -1;
}

NERF_API int nerf_mlp_pack_bf16(const nerf_net_t *net, const float *params, void *packed, nerf_stream_t stream) {
    NERF_REQUIRE(params && packed, "nerf_mlp_pack_bf16: null pointer");
    Params P;
    if (int rc = nerf::fused_net(net, P.net, "nerf_mlp_pack_bf16")) return rc;
    P.set(params);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(1024), dim3(256), 0, nerf::as_stream(stream), P,
                       reinterpret_cast<char *>(packed));
    return nerf::check_launch("nerf_mlp_pack_bf16");
}

NERF_API int64_t nerf_mlp_packed_f16x2_bytes(const nerf_net_t *net) {
    mlp::Net n;
    return nerf::f16x2_net(net, n, "nerf_mlp_packed_f16x2_bytes") == NERF_OK ? mlp::f2_layout(n.e_pos, n.e_dir).packed_bytes() : -1;
}

NERF_API int nerf_mlp_pack_f16x2(const nerf_net_t *net, const float *params, void *packed, nerf_stream_t stream) {
    NERF_REQUIRE(params && packed, "nerf_mlp_pack_f16x2: null pointer");
    Params P;
    if (int rc = nerf::f16x2_net(net, P.net, "nerf_mlp_pack_f16x2")) return rc;
    P.set(params);
    hipLaunchKernelGGL(f16x2_absmax_kernel, dim3(F2_SLICES, 10), dim3(256), 0, nerf::as_stream(stream), P,
                       reinterpret_cast<float *>(packed));
    hipLaunchKernelGGL(pack_f16x2_kernel, dim3(1024), dim3(256), 0, nerf::as_stream(stream), P,
                       reinterpret_cast<char *>(packed));
    return nerf::check_launch("nerf_mlp_pack_f16x2");
}

NERF_API int64_t nerf_mlp_packed_bytes(const nerf_net_t *net) {
    mlp::Net n;
    return nerf::fused_net(net, n, "nerf_mlp_packed_bytes") == NERF_OK ? mlp::PACKED_BYTES : -1;
}

NERF_API int nerf_mlp_pack(const nerf_net_t *net, const float *params, void *packed, nerf_stream_t stream) {
    NERF_REQUIRE(params && packed, "nerf_mlp_pack: null pointer");
    Params P;
    if (int rc = nerf::fused_net(net, P.net, "nerf_mlp_pack")) return rc;
    P.set(params);
    hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, nerf::as_stream(stream), P,
                       reinterpret_cast<float *>(packed));
    return nerf::check_launch("nerf_mlp_pack");
}
