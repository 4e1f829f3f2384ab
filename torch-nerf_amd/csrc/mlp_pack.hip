// Re-tiles the flat NeRF parameter blob (state_dict order) into the stream the fused
// MLP kernels DMA into LDS: const block + 78 forward chunks + 68 transposed chunks
// (layout: mlp_layout.h).  2.4 MB in, 4.8 MB out; runs once per parameter update.
#include "common.h"
#include "mlp_layout.h"

namespace {

using namespace mlp;

__device__ __forceinline__ float weight(const float *P, int layer, int n, int k) {
    return P[w_offset(layer) + (int64_t)n * DIMS[layer].in + k];
}

__device__ float forward_chunk_value(const float *P, int ci, int n, int kk) {
    if (ci < CH_TRUNK1) {
        const int k = 32 * ci + kk;
        return k < E_POS ? weight(P, 0, n, k) : 0.0f;
    }
    if (ci < CH_FC5_ENC) {
        const int l = 1 + (ci - CH_TRUNK1) / 8, kb = (ci - CH_TRUNK1) % 8;
        return weight(P, l, n, 32 * kb + kk);
    }
    if (ci < CH_FC5) {
        const int k = 32 * (ci - CH_FC5_ENC) + kk;
        return k < E_POS ? weight(P, 5, n, k) : 0.0f;
    }
    if (ci < CH_TRUNK6) return weight(P, 5, n, E_POS + 32 * (ci - CH_FC5) + kk);
    if (ci < CH_FC8) {
        const int l = 6 + (ci - CH_TRUNK6) / 8, kb = (ci - CH_TRUNK6) % 8;
        return weight(P, l, n, 32 * kb + kk);
    }
    if (ci < CH_FC9) return weight(P, 8, n + 1, 32 * (ci - CH_FC8) + kk);
    if (n >= HALF) return 0.0f;
    if (ci < CH_FC9 + 8) return weight(P, 9, n, 32 * (ci - CH_FC9) + kk);
    if (ci == CH_FC9 + 8) return kk < E_DIR ? weight(P, 9, n, FEAT + kk) : 0.0f;
    return 0.0f;  // filler chunk (pairs)
}

// chunk of W^T: image row m = INPUT feature, k-group = 32 consecutive OUTPUT features
__device__ float backward_chunk_value(const float *P, int ci, int m, int kk) {
    if (ci < BW_FC8T) return weight(P, 9, 32 * (ci - BW_FC9T) + kk, m);
    if (ci < BW_FC7T) return weight(P, 8, 1 + 32 * (ci - BW_FC8T) + kk, m);
    if (ci < BW_FC5T) {
        const int l = 7 - (ci - BW_FC7T) / 8, cb = (ci - BW_FC7T) % 8;
        return weight(P, l, 32 * cb + kk, m);
    }
    if (ci < BW_FC4T) return weight(P, 5, 32 * (ci - BW_FC5T) + kk, E_POS + m);
    const int l = 4 - (ci - BW_FC4T) / 8, cb = (ci - BW_FC4T) % 8;
    return weight(P, l, 32 * cb + kk, m);
}

__device__ float const_block_value(const float *P, int e) {
    if (e < CB_BIAS8) return P[b_offset(e / 256) + (e % 256)];
    if (e < CB_BIAS9) return P[b_offset(8) + 1 + (e - CB_BIAS8)];
    if (e < CB_W8ROW0) return P[b_offset(9) + (e - CB_BIAS9)];
    if (e < CB_WOUT) return weight(P, 8, 0, e - CB_W8ROW0);
    if (e < CB_SCALARS) return P[w_offset(10) + (e - CB_WOUT)];
    if (e == CB_SCALARS) return P[b_offset(8)];
    if (e < CB_SCALARS + 4) return P[b_offset(10) + (e - CB_SCALARS - 1)];
    return 0.0f;
}

__global__ void pack_kernel(const float *__restrict__ P, float *__restrict__ out) {
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
__device__ float bf16_stream_value(const float *P, int sub, int c, int n, int kk) {
    if (sub == 0 || sub == 17) {                       // encoded position into fc_in / fc_5 (pos first, nerf.py:108)
        const int k = 32 * c + kk;
        return k < E_POS ? weight(P, sub == 0 ? 0 : 5, n, k) : 0.0f;
    }
    if (sub <= 16) return weight(P, 1 + (sub - 1) / 4, n, 32 * (2 * ((sub - 1) % 4) + c) + kk);
    if (sub <= 21) return weight(P, 5, n, E_POS + 32 * (2 * (sub - 18) + c) + kk);
    if (sub <= 29) return weight(P, 6 + (sub - 22) / 4, n, 32 * (2 * ((sub - 22) % 4) + c) + kk);
    return weight(P, 8, n + 1, 32 * (2 * (sub - 30) + c) + kk);   // subs 30..33: fc_8 rows 1..256
}
// fc_9 (subs 34..36): 128-row chunks, three per sub-step; chunk index ck = 3 (sub - 34) + c is the k-block, 8 = direction
__device__ float bf16_fc9_value(const float *P, int ck, int n, int kk) {
    if (ck < 8) return weight(P, 9, n, 32 * ck + kk);
    return (ck == 8 && kk < E_DIR) ? weight(P, 9, n, FEAT + kk) : 0.0f;
}

__global__ void pack_bf16_kernel(const float *__restrict__ P, char *__restrict__ out) {
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

}  // namespace

NERF_API int64_t nerf_mlp_packed_bf16_bytes(void) { return mlp::B16_PACKED_BYTES; }

NERF_API int nerf_mlp_pack_bf16(const float *params, void *packed, nerf_stream_t stream) {
    NERF_REQUIRE(params && packed, "nerf_mlp_pack_bf16: null pointer");
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(1024), dim3(256), 0, nerf::as_stream(stream), params,
                       reinterpret_cast<char *>(packed));
    return nerf::check_launch("nerf_mlp_pack_bf16");
}

NERF_API int64_t nerf_mlp_param_count(void) { return mlp::PARAM_COUNT; }
NERF_API int64_t nerf_mlp_packed_bytes(void) { return mlp::PACKED_BYTES; }

NERF_API int nerf_mlp_pack(const float *params, void *packed, nerf_stream_t stream) {
    NERF_REQUIRE(params && packed, "nerf_mlp_pack: null pointer");
    hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, nerf::as_stream(stream), params,
                       reinterpret_cast<float *>(packed));
    return nerf::check_launch("nerf_mlp_pack");
}
