// a10 + a13 for ANY NeRF(pos_dim, view_dir_dim, feat_dim) (R/network/nerf.py:24-63): the layer-by-layer family.
//
// The register-resident kernels (mlp_forward.hip, mlp_backward.hip) are built around feat_dim = 256 and inputs of at
// most 64 / 32 encoded features.  Every other network the reference's constructor accepts -- other feat_dim, wider
// encodings (coord_encode_level > 10), any encoder output width -- runs here: one fp32-MFMA GEMM launch per layer with
// the activations in HBM, the way the reference's eager path does it (nerf.py:102-119), minus its extra passes:
// bias, ReLU / sigmoid, the two torch.cat (:108, :116) and the ReLU masks of the backward are fused into the GEMMs.
// Correct first, reasonably fast second: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), 64..128 x 64..128 x 16 LDS
// tiles, register prefetch of the next k-tile; no attempt at the 0.9-of-peak of the fused family.
//
// One kernel, three roles (all "C[i,j] = sum_k A(i,k) B(k,j)" with run-time strides):
//   forward   Y[m,n]  = act(sum_k X[m,k] W[n,k] + b[n])     k runs over ONE or TWO concatenated inputs (torch.cat)
//   dX        G'[m,k] = (sum_n G[m,n] W[n,k]) . [H[m,k] > 0]
//   dW, db    dW[n,k] = sum_m G[m,n] X[m,k], db[n] = sum_m G[m,n]: the sample axis is the reduction; it is cut into a
//             FIXED number of slices (grid.z) whose partial tiles a second kernel adds in a fixed order -- no atomics,
//             bit-reproducible gradients.  db rides along as one extra all-ones column of X.
// Backward also returns autograd's gradients w.r.t. the (encoded) inputs `pos` / `view_dir` when asked.
#include "common.h"
#include "net.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TI = 64, TJ = 64, TK = 16;   // base tile (x WI, x WJ); LDS rows are padded by one float

struct GemmArgs {
    // A operand, element (i, k): segment 1 for k < K1, segment 2 for K1 <= k < K1 + K2 (torch.cat along k)
    const float *A1; int64_t a1_si, a1_sk; int64_t K1;
    const float *A2; int64_t a2_si, a2_sk; int64_t K2;
    // B operand, element (k, j) at B[k * b_sk + j * b_sj]; column j == ones_col reads as 1.0 (bias-gradient column)
    const float *B; int64_t b_sk, b_sj;
    int I, J;                 // output extent (J counts the ones column if there is one)
    int ones_col;             // -1: none
    int a_kfast, b_kfast;     // which index is contiguous in memory (thread -> element mapping of the tile loads)
    // split of the reduction over grid.z: slice z covers k in [z * k_chunk, (z+1) * k_chunk); writes C + z * c_slice
    int64_t k_chunk, c_slice;
    // epilogue
    float *C; int64_t c_si;   // element (i, j) at C[i * c_si + j]
    const float *bias;        // per j, or null
    int act;                  // 0 none, 1 relu, 2 sigmoid
    const float *mask; int64_t mask_si;   // multiply by (mask[i * mask_si + j] > 0), or null
    int accumulate;           // C += result
};

__device__ __forceinline__ float load_a(const GemmArgs &g, int64_t i, int64_t k, int64_t k_end) {
    if (i >= g.I || k >= k_end) return 0.0f;
    return k < g.K1 ? g.A1[i * g.a1_si + k * g.a1_sk] : g.A2[i * g.a2_si + (k - g.K1) * g.a2_sk];
}
__device__ __forceinline__ float load_b(const GemmArgs &g, int64_t k, int j, int64_t k_end) {
    if (j >= g.J || k >= k_end) return 0.0f;
    return j == g.ones_col ? 1.0f : g.B[k * g.b_sk + (int64_t)j * g.b_sj];
}

// Workgroup tile (64 WI) x (64 WJ): 2 x 2 wavefronts, each (32 WI) x (32 WJ) = WI * WJ accumulator blocks.  The wider
// tiles halve the LDS reads and the staging work per MFMA (a 64 x 64 tile reads two fragments per MFMA, a 128 x 128
// tile one); the launcher picks the widest tile the output extent fills.
template <int WI, int WJ>
__global__ __launch_bounds__(256) void layered_gemm_kernel(const GemmArgs g) {
    constexpr int TI_ = TI * WI, TJ_ = TJ * WJ;
    __shared__ float As[TK][TI_ + 1], Bs[TK][TJ_ + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int64_t i0 = (int64_t)blockIdx.x * TI_;
    const int j0 = blockIdx.y * TJ_;
    const int64_t K = g.K1 + g.K2;
    const int64_t k_begin = (int64_t)blockIdx.z * g.k_chunk;
    const int64_t k_end = k_begin + g.k_chunk < K ? k_begin + g.k_chunk : K;

    // thread -> (row, k) of the 4 WI (4 WJ) tile elements it stages of the A (B) operand
    constexpr int NA = 4 * WI, NB = 4 * WJ;
    int ar[NA], ak[NA], br[NB], bk[NB];
#pragma unroll
    for (int it = 0; it < NA; ++it) {
        if (g.a_kfast) { ak[it] = tid & 15; ar[it] = (tid >> 4) + 16 * it; }
        else           { ar[it] = tid % TI_; ak[it] = tid / TI_ + (256 / TI_) * it; }
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
        if (g.b_kfast) { bk[it] = tid & 15; br[it] = (tid >> 4) + 16 * it; }
        else           { br[it] = tid % TJ_; bk[it] = tid / TJ_ + (256 / TJ_) * it; }
    }
    f32x16 acc[WI][WJ];
#pragma unroll
    for (int a = 0; a < WI; ++a)
#pragma unroll
        for (int b = 0; b < WJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    float ra[NA], rb[NB];
    auto fetch = [&](int64_t kt) {
#pragma unroll
        for (int it = 0; it < NA; ++it) ra[it] = load_a(g, i0 + ar[it], kt + ak[it], k_end);
#pragma unroll
        for (int it = 0; it < NB; ++it) rb[it] = load_b(g, kt + bk[it], j0 + br[it], k_end);
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int64_t kt = k_begin; kt < k_end; kt += TK) {
#pragma unroll
        for (int it = 0; it < NA; ++it) As[ak[it]][ar[it]] = ra[it];
#pragma unroll
        for (int it = 0; it < NB; ++it) Bs[bk[it]][br[it]] = rb[it];
        __syncthreads();
        if (kt + TK < k_end) fetch(kt + TK);   // the next tile's loads fly under this tile's MFMAs
#pragma unroll
        for (int kk = 0; kk < TK / 2; ++kk) {
            float fa[WI], fb[WJ];
#pragma unroll
            for (int a = 0; a < WI; ++a) fa[a] = As[2 * kk + (lane >> 5)][(wi * WI + a) * 32 + (lane & 31)];
#pragma unroll
            for (int b = 0; b < WJ; ++b) fb[b] = Bs[2 * kk + (lane >> 5)][(wj * WJ + b) * 32 + (lane & 31)];
#pragma unroll
            for (int a = 0; a < WI; ++a)
#pragma unroll
                for (int b = 0; b < WJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    // D fragment: register r of lane l is row (r&3) + 8 (r>>2) + 4 (l>>5), column l&31 of a 32 x 32 block
    float *C = g.C + (int64_t)blockIdx.z * g.c_slice;
#pragma unroll
    for (int b = 0; b < WJ; ++b) {
        const int j = j0 + (wj * WJ + b) * 32 + (lane & 31);
        if (j >= g.J) continue;
        const float bj = g.bias ? g.bias[j] : 0.0f;
#pragma unroll
        for (int a = 0; a < WI; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t i = i0 + (wi * WI + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (i >= g.I) continue;
                float v = acc[a][b][r] + bj;
                if (g.act == 1) v = v > 0.0f ? v : 0.0f;
                else if (g.act == 2) v = 1.0f / (1.0f + expf(-v));
                if (g.mask) v = g.mask[i * g.mask_si + j] > 0.0f ? v : 0.0f;
                float *dst = C + i * g.c_si + j;
                *dst = g.accumulate ? *dst + v : v;
            }
    }
}

// dst[i * w_si + j] = sum_z partial[z][i][j] (j < Jw), bias[i] = sum_z partial[z][i][Jw] (if bias), slices in order
__global__ void layered_reduce_kernel(const float *__restrict__ partial, int slices, int I, int J, int Jw,
                                      float *__restrict__ w, int64_t w_si, float *__restrict__ bias) {
    const int64_t total = (int64_t)I * J;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / J), j = (int)(e % J);
        float s = 0.0f;
        for (int z = 0; z < slices; ++z) s += partial[(int64_t)z * total + e];
        if (j < Jw) w[i * w_si + j] = s;
        else if (bias) bias[i] = s;
    }
}

// sigma = relu(y8[:, 0])   (nerf.py:115)
__global__ void layered_sigma_kernel(const float *__restrict__ y8, int64_t ld, int64_t M, float *__restrict__ sigma) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) {
        const float v = y8[m * ld];
        sigma[m] = v > 0.0f ? v : 0.0f;
    }
}

// reverse of the two output non-linearities (nerf.py:115, :119): g10 = g_rgb * rgb * (1 - rgb); gy8[:, 0] = g_sigma . [sigma > 0]
__global__ void layered_heads_bwd_kernel(const float *__restrict__ sigma, const float *__restrict__ rgb,
                                         const float *__restrict__ g_sigma, const float *__restrict__ g_rgb, int64_t M,
                                         float *__restrict__ g10, float *__restrict__ gy8, int64_t ld8) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float y = rgb[3 * m + c];
        g10[3 * m + c] = g_rgb[3 * m + c] * y * (1.0f - y);
    }
    gy8[m * ld8] = sigma[m] > 0.0f ? g_sigma[m] : 0.0f;
}

struct Layout {          // flat parameter blob, state_dict order
    int E_p, E_d, F, H;
    int in[11], out[11];
    int64_t w[11], b[11], total;
    explicit Layout(const nerf_net_t &d) : E_p(d.pos_dim), E_d(d.view_dir_dim), F(d.feat_dim), H(d.feat_dim / 2) {
        const int ins[11] = {E_p, F, F, F, F, F + E_p, F, F, F, F + E_d, H};
        const int outs[11] = {F, F, F, F, F, F, F, F, F + 1, H, 3};
        int64_t off = 0;
        for (int l = 0; l < 11; ++l) {
            in[l] = ins[l]; out[l] = outs[l];
            w[l] = off; off += (int64_t)ins[l] * outs[l];
            b[l] = off; off += outs[l];
        }
        total = off;
    }
    int64_t record_floats_per_row() const { return 8 * (int64_t)F + (F + 1) + H; }   // h0..h7, y8, h9
};

struct Record {          // row-major planes of `rows` rows
    float *h[8], *y8, *h9;
    Record(float *base, int64_t rows, const Layout &L) {
        for (int l = 0; l < 8; ++l) h[l] = base + rows * (int64_t)L.F * l;
        y8 = base + rows * (int64_t)L.F * 8;
        h9 = y8 + rows * (int64_t)(L.F + 1);
    }
};

GemmArgs blank() {
    GemmArgs g = {};
    g.ones_col = -1;
    return g;
}

// Tile shape.  The kernel is templated on 64- or 128-wide tiles per direction; the 128-wide ones (two accumulator
// blocks per wavefront and direction, half the LDS reads per MFMA) were measured and are SLOWER -- 60 / 43 instead of
// 61.5 / 53 TFLOP/s forward / backward at feat_dim 256 -- because the kernel is bound by its bounds-checked scalar
// staging loads, which the 64 x 64 tile hides behind four workgroups per CU.  So: 64 x 64 everywhere.
#ifdef X_LAYERED_WIDE
inline int tile_i(int I) { return I >= 128 ? 128 : 64; }
inline int tile_j(int J) { return J >= 128 ? 128 : 64; }
#else
inline int tile_i(int) { return 64; }
inline int tile_j(int) { return 64; }
#endif

int launch(const GemmArgs &g, int slices, hipStream_t s, const char *what) {
    if (g.I <= 0 || g.J <= 0) return NERF_OK;
    const int ti = tile_i(g.I), tj = tile_j(g.J);
    const dim3 grid((unsigned)((g.I + ti - 1) / ti), (unsigned)((g.J + tj - 1) / tj), (unsigned)slices);
    if (ti == 128 && tj == 128) hipLaunchKernelGGL((layered_gemm_kernel<2, 2>), grid, dim3(256), 0, s, g);
    else if (ti == 128) hipLaunchKernelGGL((layered_gemm_kernel<2, 1>), grid, dim3(256), 0, s, g);
    else if (tj == 128) hipLaunchKernelGGL((layered_gemm_kernel<1, 2>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((layered_gemm_kernel<1, 1>), grid, dim3(256), 0, s, g);
    return nerf::check_launch(what);
}

// Y[rows, N] = act(cat(X1, X2) W^T + b): nn.Linear (+ torch.cat of its two inputs) + activation
int linear_fwd(const float *P, const Layout &L, int layer, const float *X1, int64_t ld1, int K1, const float *X2,
               int64_t ld2, int K2, int64_t rows, float *Y, int64_t ldy, int act, hipStream_t s) {
    GemmArgs g = blank();
    g.A1 = X1; g.a1_si = ld1; g.a1_sk = 1; g.K1 = K1;
    g.A2 = X2; g.a2_si = ld2; g.a2_sk = 1; g.K2 = K2;
    g.B = P + L.w[layer]; g.b_sk = 1; g.b_sj = L.in[layer];       // B(k, n) = W[n][k]
    g.I = (int)rows; g.J = L.out[layer];
    g.a_kfast = 1; g.b_kfast = 1;
    g.k_chunk = K1 + K2; g.c_slice = 0;
    g.C = Y; g.c_si = ldy; g.bias = P + L.b[layer]; g.act = act;
    return launch(g, 1, s, "nerf_mlp_layered_forward: layer");
}

// dX[rows, K] (+)= (G[rows, N] W[:, col0 : col0 + K]) . [mask > 0]
int linear_dx(const float *P, const Layout &L, int layer, const float *G, int64_t ldg, int64_t rows, int col0, int K,
              float *dX, int64_t ldx, const float *mask, int64_t ldm, int accumulate, hipStream_t s) {
    GemmArgs g = blank();
    g.A1 = G; g.a1_si = ldg; g.a1_sk = 1; g.K1 = L.out[layer];    // reduction over the layer's outputs
    g.B = P + L.w[layer] + col0; g.b_sk = L.in[layer]; g.b_sj = 1;  // B(n, k) = W[n][col0 + k]
    g.I = (int)rows; g.J = K;
    g.a_kfast = 1; g.b_kfast = 0;
    g.k_chunk = g.K1; g.c_slice = 0;
    g.C = dX; g.c_si = ldx; g.mask = mask; g.mask_si = ldm; g.accumulate = accumulate;
    return launch(g, 1, s, "nerf_mlp_layered_backward: dX");
}

// fixed slicing of the sample axis (a function of M and the tile grid only: never of the device)
int dw_slices(int64_t M, int I, int J) {
    const int64_t tiles = (int64_t)((I + tile_i(I) - 1) / tile_i(I)) * ((J + tile_j(J) - 1) / tile_j(J));
    int64_t want = (1024 + tiles - 1) / tiles;
    const int64_t most = (M + 511) / 512;          // at least 512 rows per slice
    if (want > most) want = most;
    if (want > 256) want = 256;
    return (int)(want < 1 ? 1 : want);
}

// dW[:, col0 : col0 + K] = G^T X (and db = column sums of G when with_bias), through `partial`
int linear_dw(float *GP, const Layout &L, int layer, const float *G, int64_t ldg, const float *X, int64_t ldx,
              int64_t M, int col0, int K, bool with_bias, float *partial, hipStream_t s) {
    const int I = L.out[layer], J = K + (with_bias ? 1 : 0);
    const int slices = dw_slices(M, I, J);
    GemmArgs g = blank();
    g.A1 = G; g.a1_si = 1; g.a1_sk = ldg; g.K1 = M;               // A(n, m) = G[m][n]
    g.B = X; g.b_sk = ldx; g.b_sj = 1;                            // B(m, k) = X[m][k]
    g.I = I; g.J = J; g.ones_col = with_bias ? K : -1;
    g.a_kfast = 0; g.b_kfast = 0;
    g.k_chunk = ((M + slices - 1) / slices + TK - 1) / TK * TK;
    g.c_slice = (int64_t)I * J;
    g.C = partial; g.c_si = J;
    if (int rc = launch(g, slices, s, "nerf_mlp_layered_backward: dW")) return rc;
    const int64_t total = (int64_t)I * J;
    hipLaunchKernelGGL(layered_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)),
                       dim3(256), 0, s, partial, slices, I, J, K, GP + L.w[layer] + col0, (int64_t)L.in[layer],
                       with_bias ? GP + L.b[layer] : nullptr);
    return nerf::check_launch("nerf_mlp_layered_backward: reduce");
}

int64_t align64(int64_t floats) { return (floats + 63) & ~(int64_t)63; }

int64_t partial_floats(const Layout &L) {   // the largest slices * I * J over the dW calls of the backward, for any M
    int64_t most = 0;
    auto piece = [&](int I, int J) {
        const int64_t tiles = (int64_t)((I + tile_i(I) - 1) / tile_i(I)) * ((J + tile_j(J) - 1) / tile_j(J));
        int64_t s = (1024 + tiles - 1) / tiles;
        if (s > 256) s = 256;
        if (s * I * J > most) most = s * (int64_t)I * J;
    };
    for (int l = 0; l < 11; ++l) {
        if (l == 5) { piece(L.out[l], L.E_p); piece(L.out[l], L.F + 1); }          // (the bias column counts)
        else if (l == 9) { piece(L.out[l], L.F + 1); piece(L.out[l], L.E_d); }
        else piece(L.out[l], L.in[l] + 1);
    }
    return most;
}

}  // namespace

NERF_API int64_t nerf_mlp_layered_record_bytes(const nerf_net_t *net, int64_t rows) {
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0) return -1;
    return rows <= 0 ? 0 : 4 * rows * Layout(d).record_floats_per_row();
}

NERF_API int64_t nerf_mlp_layered_workspace_bytes(const nerf_net_t *net, int64_t M) {
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0) return -1;
    if (M <= 0) return 0;
    const Layout L(d);
    // two ping-pong gradient planes (M, F+1), dY9 (M, F/2), d y10 (M, 3), partial dW tiles
    return 4 * (2 * align64(M * (int64_t)(L.F + 1)) + align64(M * (int64_t)L.H) + align64(M * 3) + align64(partial_floats(L)));
}

NERF_API int nerf_mlp_layered_forward(const nerf_net_t *net, const float *params, const float *pos,
                                      const float *view_dir, int64_t M, float *sigma, float *rgb, void *record,
                                      int64_t record_rows, nerf_stream_t stream) {
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0) return NERF_ERR_ARG;
    NERF_REQUIRE(M >= 0, "nerf_mlp_layered_forward: negative M");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(params && pos && view_dir && sigma && rgb && record && record_rows > 0,
                 "nerf_mlp_layered_forward: null pointer");
    NERF_REQUIRE(M < (int64_t)1 << 31, "nerf_mlp_layered_forward: more than 2^31 samples per call");
    const Layout L(d);
    hipStream_t s = nerf::as_stream(stream);
    const int64_t chunk = record_rows < M ? record_rows : M;
    const Record R(static_cast<float *>(record), chunk, L);
    const int F = L.F, E_p = L.E_p, E_d = L.E_d;
    for (int64_t r0 = 0; r0 < M; r0 += chunk) {
        const int64_t rows = M - r0 < chunk ? M - r0 : chunk;
        const float *x = pos + r0 * E_p, *v = view_dir + r0 * E_d;
        int rc;
        // nerf.py:102-106
        if ((rc = linear_fwd(params, L, 0, x, E_p, E_p, nullptr, 0, 0, rows, R.h[0], F, 1, s))) return rc;
        for (int l = 1; l <= 4; ++l)
            if ((rc = linear_fwd(params, L, l, R.h[l - 1], F, F, nullptr, 0, 0, rows, R.h[l], F, 1, s))) return rc;
        // :108-110  cat([pos, x]) -- pos first
        if ((rc = linear_fwd(params, L, 5, x, E_p, E_p, R.h[4], F, F, rows, R.h[5], F, 1, s))) return rc;
        for (int l = 6; l <= 7; ++l)
            if ((rc = linear_fwd(params, L, l, R.h[l - 1], F, F, nullptr, 0, 0, rows, R.h[l], F, 1, s))) return rc;
        // :113-115  fc_8 has no ReLU; sigma = relu(x[:, 0])
        if ((rc = linear_fwd(params, L, 8, R.h[7], F, F, nullptr, 0, 0, rows, R.y8, F + 1, 0, s))) return rc;
        hipLaunchKernelGGL(layered_sigma_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, R.y8,
                           (int64_t)(F + 1), rows, sigma + r0);
        if ((rc = nerf::check_launch("nerf_mlp_layered_forward: sigma"))) return rc;
        // :116-119  cat([x[:, 1:], view_dir]) -- features first; rgb = sigmoid(fc_out(.))
        if ((rc = linear_fwd(params, L, 9, R.y8 + 1, F + 1, F, v, E_d, E_d, rows, R.h9, L.H, 1, s))) return rc;
        if ((rc = linear_fwd(params, L, 10, R.h9, L.H, L.H, nullptr, 0, 0, rows, rgb + 3 * r0, 3, 2, s))) return rc;
    }
    return NERF_OK;
}

NERF_API int nerf_mlp_layered_backward(const nerf_net_t *net, const float *params, const float *pos,
                                       const float *view_dir, int64_t M, const float *sigma, const float *rgb,
                                       const void *record, const float *g_sigma, const float *g_rgb,
                                       float *g_params, float *g_pos, float *g_view_dir, void *workspace,
                                       nerf_stream_t stream) {
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0) return NERF_ERR_ARG;
    NERF_REQUIRE(M >= 0, "nerf_mlp_layered_backward: negative M");
    NERF_REQUIRE(g_params, "nerf_mlp_layered_backward: null g_params");
    const Layout L(d);
    hipStream_t s = nerf::as_stream(stream);
    if (M == 0) {
        if (hipMemsetAsync(g_params, 0, sizeof(float) * L.total, s) != hipSuccess)
            return nerf::check_launch("nerf_mlp_layered_backward: memset");
        return NERF_OK;
    }
    NERF_REQUIRE(params && pos && view_dir && sigma && rgb && record && g_sigma && g_rgb && workspace,
                 "nerf_mlp_layered_backward: null pointer");
    NERF_REQUIRE(M < (int64_t)1 << 31, "nerf_mlp_layered_backward: more than 2^31 samples per call");
    const int F = L.F, H = L.H, E_p = L.E_p, E_d = L.E_d, LD = F + 1;
    const Record R(static_cast<float *>(const_cast<void *>(record)), M, L);
    float *ga = static_cast<float *>(workspace);
    float *gb = ga + align64(M * (int64_t)LD);
    float *g9 = gb + align64(M * (int64_t)LD);
    float *g10 = g9 + align64(M * (int64_t)H);
    float *partial = g10 + align64(M * 3);
    int rc;
    // sigmoid / relu of the two heads; gb[:, 0] = d y8[:, 0]
    hipLaunchKernelGGL(layered_heads_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, sigma, rgb, g_sigma,
                       g_rgb, M, g10, gb, (int64_t)LD);
    if ((rc = nerf::check_launch("nerf_mlp_layered_backward: heads"))) return rc;
    // fc_out (:119)
    if ((rc = linear_dw(g_params, L, 10, g10, 3, R.h9, H, M, 0, H, true, partial, s))) return rc;
    if ((rc = linear_dx(params, L, 10, g10, 3, M, 0, H, g9, H, R.h9, H, 0, s))) return rc;           // . [h9 > 0]
    // fc_9 on cat([y8[:, 1:], view_dir]) (:116-118)
    if ((rc = linear_dw(g_params, L, 9, g9, H, R.y8 + 1, LD, M, 0, F, true, partial, s))) return rc;
    if ((rc = linear_dw(g_params, L, 9, g9, H, view_dir, E_d, M, F, E_d, false, partial, s))) return rc;
    if ((rc = linear_dx(params, L, 9, g9, H, M, 0, F, gb + 1, LD, nullptr, 0, 0, s))) return rc;     // d y8[:, 1:] (no ReLU)
    if (g_view_dir && (rc = linear_dx(params, L, 9, g9, H, M, F, E_d, g_view_dir, E_d, nullptr, 0, 0, s))) return rc;
    // fc_8 (:113): gb = d y8 (M, F+1)
    if ((rc = linear_dw(g_params, L, 8, gb, LD, R.h[7], F, M, 0, F, true, partial, s))) return rc;
    if ((rc = linear_dx(params, L, 8, gb, LD, M, 0, F, ga, LD, R.h[7], F, 0, s))) return rc;         // . [h7 > 0]
    // fc_7, fc_6
    float *cur = ga, *nxt = gb;
    for (int l = 7; l >= 6; --l) {
        if ((rc = linear_dw(g_params, L, l, cur, LD, R.h[l - 1], F, M, 0, F, true, partial, s))) return rc;
        if ((rc = linear_dx(params, L, l, cur, LD, M, 0, F, nxt, LD, R.h[l - 1], F, 0, s))) return rc;
        float *t = cur; cur = nxt; nxt = t;
    }
    // fc_5 on cat([pos, h4]) (:108-110)
    if ((rc = linear_dw(g_params, L, 5, cur, LD, pos, E_p, M, 0, E_p, false, partial, s))) return rc;
    if ((rc = linear_dw(g_params, L, 5, cur, LD, R.h[4], F, M, E_p, F, true, partial, s))) return rc;
    if (g_pos && (rc = linear_dx(params, L, 5, cur, LD, M, 0, E_p, g_pos, E_p, nullptr, 0, 0, s))) return rc;
    if ((rc = linear_dx(params, L, 5, cur, LD, M, E_p, F, nxt, LD, R.h[4], F, 0, s))) return rc;     // . [h4 > 0]
    { float *t = cur; cur = nxt; nxt = t; }
    // fc_4 .. fc_1
    for (int l = 4; l >= 1; --l) {
        if ((rc = linear_dw(g_params, L, l, cur, LD, R.h[l - 1], F, M, 0, F, true, partial, s))) return rc;
        if ((rc = linear_dx(params, L, l, cur, LD, M, 0, F, nxt, LD, R.h[l - 1], F, 0, s))) return rc;
        float *t = cur; cur = nxt; nxt = t;
    }
    // fc_in (:102); autograd ADDS its contribution to `pos` to the skip connection's
    if ((rc = linear_dw(g_params, L, 0, cur, LD, pos, E_p, M, 0, E_p, true, partial, s))) return rc;
    if (g_pos && (rc = linear_dx(params, L, 0, cur, LD, M, 0, E_p, g_pos, E_p, nullptr, 0, 1, s))) return rc;
    return NERF_OK;
}
