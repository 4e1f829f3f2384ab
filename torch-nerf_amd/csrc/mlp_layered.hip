// a10 + a13 for ANY NeRF(pos_dim, view_dir_dim, feat_dim) (R/network/nerf.py:24-63): the "layered" family.
//
// The register-resident kernels (mlp_forward.hip, mlp_backward.hip) carry a sample's 256 activations through the
// whole network in registers.  A wider network (feat_dim 512: 512 inputs + 512 outputs per layer) does not fit, so
// this family PARKS the activations of a tile in HBM planes between layers -- and keeps everything else of the
// fused design:
//  * one persistent launch per forward (and one per reverse chain); workgroup = 4 wavefronts x 32 samples; a workgroup
//    takes a 128-sample tile through ALL layers itself, so no workgroup ever waits for another one: a lane reads back,
//    one layer later, exactly the 16-byte groups it stored (the tile-fragment plane layout of mlp_layout.h: the D
//    fragment a lane holds IS the B fragment it needs next) -- L2-hot, coalesced 1-KiB loads / stores
//  * transposed GEMM Y^T = W X^T on v_mfma_f32_32x32x2_f32, weights = A operand streamed L2 -> LDS ring by LDS-DMA
//    (mlp_device.h: Pipe), pre-packed once per call into zero-padded slot-major 64-KiB pairs: NO bounds checks, NO
//    branches and no address arithmetic in the k-loop (the first version of this file: per-element bounds-checked
//    scalar staging loads, 64 x 64 x 16 tiles, one launch per layer: 0.18-0.46 of the MFMA peak)
//  * a layer is cut into PASSES of <= 8 output blocks (256 features = 128 accumulator registers); the B operand of a
//    pass streams through registers one pair (64 KiB of weights) ahead (compiler-counted loads; in the eight-block
//    passes they ride between the MFMA groups of the pair in front of them)
//  * bias = initial accumulator, ReLU / the ReLU mask of the reverse chain in the epilogue, both torch.cat
//    (nerf.py:108, :116) = a pass with two source planes; the density row and fc_out (+ sigmoid) are vector side jobs
//  * dW = the dW GEMM kernel of mlp_backward.hip over the same planes (nerf::run_dw_items), thin rows by a vector kernel
// Padded widths (multiples of 32) carry exact zeros, so they add nothing to any sum.
//
// The pass programs are pure functions of the widths (fwd_pass / dx_pass, __host__ __device__): the persistent kernels
// tabulate theirs in LDS once per launch, the pack kernel evaluates them per slot.
#include <type_traits>
#include <vector>

#include "mlp_device.h"
#include "net.h"

namespace nerf {
// mlp_backward.hip: the dW GEMMs of the fused family over arbitrary TF planes
struct DwItem {
    const float *a_plane; int a_width, a_fb0, a_blocks;   // dY plane (floats/sample), first block, blocks wanted (<= 8)
    const float *x_plane; int x_width, x_fb0, x_blocks;   // X plane, window of <= 8 blocks
    float *w_dst; int ld;                                 // dW[(a_fb0*32 + n) * ld + x_fb0*32 + k] destination (row n, col k)
    int rows_valid, cols_valid;                           // of the window
    float *b_dst;                                         // bias gradient for the window's rows, or null
    int row0;                                             // destination rows in front of the window that w_dst / b_dst do not include
    int side;                                             // 0 | 2 density row of fc_8 | 4 fc_out.weight: summed beside this item's GEMM
};
struct DwSide { const float *h9, *dsig, *gy; float *wout; };   // planes of the side jobs, fc_out.weight in the flat gradient
int run_dw_items(const std::vector<DwItem> &items, int64_t M, void *scratch, int64_t scratch_bytes, hipStream_t s,
                 const DwSide *side);
int64_t dw_items_scratch_bytes(int n_items);
int64_t dw_items_needed_bytes(const std::vector<DwItem> &items, int64_t M, int cus);                      // host only
void dw_item_read_extent(const DwItem &it, int64_t M, int64_t *a_floats, int64_t *x_floats);            // host only
}  // namespace nerf

#ifdef X_LAYERED_TIMELINE   // scripts/timeline_layered.py: where workgroup 0 / wavefront 0 of the general kernel spends its cycles
__device__ unsigned long long g_lt[8];   // drain, operand issue + init, acquire waits, MFMA loops, epilogue, between passes, passes
#define LT_NOW() __builtin_readcyclecounter()
#define LT_ADD(k, v) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_lt[k] += (v); } while (0)
#else
#define LT_NOW() 0ull
#define LT_ADD(k, v) do { (void)(v); } while (0)
#endif

namespace {

using namespace mlp;

constexpr int ROUND32(int x) { return (x + 31) & ~31; }
// rows of every plane of this family: whole 256-sample tiles (the narrow kernels walk 64 samples per wavefront)
__host__ __device__ constexpr int64_t lrows(int64_t M) { return (M + 255) / 256 * 256; }

// ---------------------------------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------------------------------
struct Dims {
    int E_p, E_d, F, H;          // nerf.py:49-59
    int Pp, Dp, Fp, Hp;          // plane widths: padded to whole 32-feature blocks
    int64_t w[11], b[11], total; // flat parameter blob, state_dict order
    int in[11], out[11];
    __host__ __device__ int mw() const { return (Fp / 32 + 1) / 2; }               // ReLU-mask dwords per (sample, lane half) and layer
    __host__ __device__ int recw() const { return Pp + Dp + 9 * Fp + 18 * mw() + Hp; }   // record floats per sample
    __host__ __device__ int gradw() const { return 9 * Fp + Hp + 8 + Pp + Dp; }     // gradient planes, floats per sample
    // record planes, float offset per sample (x MP)
    __host__ __device__ int r_pe() const { return 0; }
    __host__ __device__ int r_de() const { return Pp; }
    __host__ __device__ int r_h(int l) const { return Pp + Dp + l * Fp; }           // h0..h7 (l = 8: y8)
    // ReLU decisions as BIT planes (l = 0..7: h_l > 0; 8: h9 > 0), what the reverse chain reads instead of the activation
    // planes: dword (2 m + h) mw + fb / 2 of plane l holds, for lane half h of sample m, bit 16 (fb & 1) + r = register r
    // of feature block fb (the fused family's save_mask / masked layout, mlp_device.h)
    __host__ __device__ int r_mask(int l) const { return Pp + Dp + 9 * Fp + 2 * mw() * l; }
    __host__ __device__ int r_h9() const { return Pp + Dp + 9 * Fp + 18 * mw(); }
    // gradient planes
    __host__ __device__ int g_dy(int l) const { return l * Fp; }                    // dY0..dY7, dY8
    __host__ __device__ int g_dy9() const { return 9 * Fp; }
    __host__ __device__ int g_gy() const { return 9 * Fp + Hp; }                    // [sample][4], row-major
    __host__ __device__ int g_dsig() const { return 9 * Fp + Hp + 4; }              // [sample] (4 reserved)
    __host__ __device__ int g_gp() const { return 9 * Fp + Hp + 8; }
    __host__ __device__ int g_gd() const { return 9 * Fp + Hp + 8 + Pp; }
    // constant block (floats): fc_8.weight[0, :], fc_out.weight, 4 scalars, then one zero-padded bias row per layer
    __host__ __device__ int c_w8row() const { return 0; }
    __host__ __device__ int c_wout() const { return Fp; }                           // 3 rows of Hp
    __host__ __device__ int c_scal() const { return Fp + 3 * Hp; }                  // fc_8.bias[0], fc_out.bias[0..2]
    __host__ __device__ int c_bias(int l) const { return Fp + 3 * Hp + 4 + l * Fp; } // l = 0..9 (fc_8: rows 1..F)
    __host__ __device__ int c_floats() const { return (Fp + 3 * Hp + 4 + 9 * Fp + Hp + 63) & ~63; }
};

Dims make_dims(const nerf_net_t &d) {
    Dims D;
    D.E_p = d.pos_dim; D.E_d = d.view_dir_dim; D.F = d.feat_dim; D.H = d.feat_dim / 2;
    D.Pp = ROUND32(D.E_p); D.Dp = ROUND32(D.E_d); D.Fp = ROUND32(D.F); D.Hp = ROUND32(D.H);
    const int ins[11] = {D.E_p, D.F, D.F, D.F, D.F, D.F + D.E_p, D.F, D.F, D.F, D.F + D.E_d, D.H};
    const int outs[11] = {D.F, D.F, D.F, D.F, D.F, D.F, D.F, D.F, D.F + 1, D.H, 3};
    int64_t off = 0;
    for (int l = 0; l < 11; ++l) {
        D.in[l] = ins[l]; D.out[l] = outs[l];
        D.w[l] = off; off += (int64_t)ins[l] * outs[l];
        D.b[l] = off; off += outs[l];
    }
    D.total = off;
    return D;
}

// One pass: NFB accumulator blocks of one layer's output, all of its k.  Planes are named by (buffer, float offset per
// sample, width); buffer 0 = the activation record, 1 = the gradient workspace.
enum { INIT_ZERO = 0, INIT_BIAS = 1, INIT_DENSITY = 2 };
enum { SIDE_NONE = 0, SIDE_DENSITY = 1, SIDE_FCOUT = 2 };
struct Pass {
    int nfb;                 // accumulator blocks computed: 2 | 4 | 8 (zero rows beyond `blocks`)
    int blocks;              // blocks that exist in the destination window
    int kb0, kb1;            // k-blocks of the two sources (torch.cat); kb1 = 0: one source
    int src_buf0, src_buf1, src_off0, src_off1, src_w0, src_w1;   // (scalars, not arrays: hipcc parks a struct's arrays in scratch)
    int dst_buf, dst_off, dst_w, dst_fb0;
    int mask_off;            // bit plane (r_mask) -- reverse chain: the one that masks the result, forward: the one the
                             // pass's ReLU decisions are written to; -1: none.  Words dst_fb0 / 2 .. of every (sample, half)
    int init, init_off;      // INIT_*: constant-block offset of the bias window / the density row
    int relu;
    int side;
    int first;               // first pass of its layer (side jobs that accumulate over passes start here)
    int last;                // last pass of its layer
    // weights: W[layer[s]] element (row, col); forward: out n -> row row0 + n, k -> col col0[s] + k;
    // transposed (reverse chain): out n -> col col0[s] + n, k -> row row0 + k
    int transposed, layer0, layer1, row0, rows_valid, col00, col01, cols_valid0, cols_valid1;
    __host__ __device__ int chunks() const { const int kpc = 8 / nfb; return (kb0 + kb1 + kpc - 1) / kpc; }
    __host__ __device__ int pairs() const { return (chunks() + 1) / 2; }
};

// every field of a Pass, in one place: the persistent kernel tabulates its program in LDS once (one lane per pass) and
// fetches a pass with one LDS read + one v_readlane per field (the scalar evaluation of fwd_pass / dx_pass cost 1.3-3.7 k
// cycles per pass: scripts/timeline_layered.py)
#define PASS_FIELDS(X) X(nfb) X(blocks) X(kb0) X(kb1) X(src_buf0) X(src_buf1) X(src_off0) X(src_off1) X(src_w0) X(src_w1) \
    X(dst_buf) X(dst_off) X(dst_w) X(dst_fb0) X(mask_off) X(init) X(init_off) X(relu) X(side) X(first) X(last) X(transposed) \
    X(layer0) X(layer1) X(row0) X(rows_valid) X(col00) X(col01) X(cols_valid0) X(cols_valid1)
constexpr int PASS_ROW = 32;           // ints per tabulated pass (30 fields: one per lane of half a wavefront)
constexpr int PASS_TABLE_MAX = 128;    // passes the LDS table holds (feat_dim <= 2048); longer programs are evaluated on the fly
static_assert(sizeof(Pass) == 30 * sizeof(int), "PASS_FIELDS must name every field of Pass");

__host__ __device__ inline int pass_nfb(int blocks) { return blocks > 4 ? 8 : blocks > 2 ? 4 : 2; }

// Forward program: layer by layer (nerf.py:102-119), each layer in passes of <= 8 output blocks.
__host__ __device__ inline int fwd_layer_passes(const Dims &D, int l) { return ((l == 9 ? D.Hp : D.Fp) / 32 + 7) / 8; }
__host__ __device__ inline int fwd_num_passes(const Dims &D) {
    int n = 0;
    for (int l = 0; l <= 9; ++l) n += fwd_layer_passes(D, l);
    return n;
}
__host__ __device__ inline Pass fwd_pass(const Dims &D, int idx) {
    int l = 0, j = idx;
    while (j >= fwd_layer_passes(D, l)) { j -= fwd_layer_passes(D, l); ++l; }
    const int nblk = (l == 9 ? D.Hp : D.Fp) / 32;
    Pass P = {};
    P.blocks = nblk - 8 * j < 8 ? nblk - 8 * j : 8;
    P.nfb = pass_nfb(P.blocks);
    P.dst_buf = 0; P.dst_fb0 = 8 * j; P.mask_off = l == 9 ? D.r_mask(8) : l == 8 ? -1 : D.r_mask(l);
    P.dst_off = l == 9 ? D.r_h9() : D.r_h(l); P.dst_w = l == 9 ? D.Hp : D.Fp;
    P.init = INIT_BIAS; P.init_off = D.c_bias(l) + 256 * j;
    P.relu = l != 8;                                          // fc_8 has no ReLU (nerf.py:113)
    P.first = j == 0; P.last = j + 1 == fwd_layer_passes(D, l);
    P.side = l == 8 ? (j == 0 ? SIDE_DENSITY : SIDE_NONE) : l == 9 ? SIDE_FCOUT : SIDE_NONE;
    P.transposed = 0; P.layer0 = P.layer1 = l;
    P.row0 = l == 8 ? 1 : 0;                                  // row 0 of fc_8 is the density row
    P.rows_valid = l == 9 ? D.H : D.F;
    P.src_buf0 = P.src_buf1 = 0;
    if (l == 0) {
        P.kb0 = D.Pp / 32; P.src_off0 = D.r_pe(); P.src_w0 = D.Pp; P.col00 = 0; P.cols_valid0 = D.E_p;
    } else if (l == 5) {   // cat([pos, x]) (:108)
        P.kb0 = D.Pp / 32; P.src_off0 = D.r_pe(); P.src_w0 = D.Pp; P.col00 = 0; P.cols_valid0 = D.E_p;
        P.kb1 = D.Fp / 32; P.src_off1 = D.r_h(4); P.src_w1 = D.Fp; P.col01 = D.E_p; P.cols_valid1 = D.F;
    } else if (l == 9) {   // cat([x[:, 1:], view_dir]) (:116)
        P.kb0 = D.Fp / 32; P.src_off0 = D.r_h(8); P.src_w0 = D.Fp; P.col00 = 0; P.cols_valid0 = D.F;
        P.kb1 = D.Dp / 32; P.src_off1 = D.r_de(); P.src_w1 = D.Dp; P.col01 = D.F; P.cols_valid1 = D.E_d;
    } else {
        P.kb0 = D.Fp / 32; P.src_off0 = D.r_h(l - 1); P.src_w0 = D.Fp; P.col00 = 0; P.cols_valid0 = D.F;
    }
    return P;
}

// Reverse chain (autograd of nerf.py:102-119), after the vector prologue that makes dY9:
//   stage 0: d y8[1:]  = W9[:, :F]^T dY9                       stage 1 (inputs): g_view_dir = W9[:, F:]^T dY9
//   stage 2: dY7 = (W8[1:]^T d y8 + W8[0] dsigma') . [h7 > 0]  stages 3, 4: dY6, dY5
//   stage 5: dY4 = (W5[:, E_p:]^T dY5) . [h4 > 0]              stages 6..9: dY3 .. dY0
//   stage 10 (inputs): g_pos = W5[:, :E_p]^T dY5 + W_in^T dY0  (autograd adds the two shares as well)
__host__ __device__ inline int dx_stage_blocks(const Dims &D, int st) { return (st == 1 ? D.Dp : st == 10 ? D.Pp : D.Fp) / 32; }
__host__ __device__ inline int dx_stage_passes(const Dims &D, int st, int inputs) {
    if ((st == 1 || st == 10) && !inputs) return 0;
    return (dx_stage_blocks(D, st) + 7) / 8;
}
__host__ __device__ inline int dx_num_passes(const Dims &D, int inputs) {
    int n = 0;
    for (int st = 0; st <= 10; ++st) n += dx_stage_passes(D, st, inputs);
    return n;
}
__host__ __device__ inline Pass dx_pass(const Dims &D, int inputs, int idx) {
    int st = 0, j = idx;
    while (j >= dx_stage_passes(D, st, inputs)) { j -= dx_stage_passes(D, st, inputs); ++st; }
    const int nblk = dx_stage_blocks(D, st);
    Pass P = {};
    P.blocks = nblk - 8 * j < 8 ? nblk - 8 * j : 8;
    P.nfb = pass_nfb(P.blocks);
    P.dst_buf = 1; P.dst_fb0 = 8 * j; P.mask_off = -1; P.init = INIT_ZERO; P.relu = 0; P.side = SIDE_NONE;
    P.first = j == 0; P.last = j + 1 == dx_stage_passes(D, st, inputs);
    P.transposed = 1; P.src_buf0 = P.src_buf1 = 1;
    if (st <= 1) {          // sources: dY9 (H outputs of fc_9)
        P.kb0 = D.Hp / 32; P.src_off0 = D.g_dy9(); P.src_w0 = D.Hp;
        P.layer0 = 9; P.row0 = 0; P.rows_valid = D.H;
        if (st == 0) { P.dst_off = D.g_dy(8); P.dst_w = D.Fp; P.col00 = 0; P.cols_valid0 = D.F; }
        else         { P.dst_off = D.g_gd(); P.dst_w = D.Dp; P.col00 = D.F; P.cols_valid0 = D.E_d; }
    } else if (st == 10) {  // g_pos: two sources, two layers
        // (dY0 first: the narrow kernels still hold it in registers when they get here)
        P.kb0 = D.Fp / 32; P.src_off0 = D.g_dy(0); P.src_w0 = D.Fp; P.layer0 = 0;
        P.kb1 = D.Fp / 32; P.src_off1 = D.g_dy(5); P.src_w1 = D.Fp; P.layer1 = 5;
        P.row0 = 0; P.rows_valid = D.F; P.col00 = P.col01 = 0; P.cols_valid0 = P.cols_valid1 = D.E_p;
        P.dst_off = D.g_gp(); P.dst_w = D.Pp;
    } else {                // stage st = 2..9: layer l = 10 - st (8 .. 1): dY(l-1) from dY(l)
        const int l = 10 - st;
        P.kb0 = D.Fp / 32; P.src_off0 = D.g_dy(l); P.src_w0 = D.Fp; P.layer0 = l;
        P.row0 = l == 8 ? 1 : 0; P.rows_valid = D.F;
        P.col00 = l == 5 ? D.E_p : 0; P.cols_valid0 = D.F;
        P.dst_off = D.g_dy(l - 1); P.dst_w = D.Fp;
        P.mask_off = D.r_mask(l - 1);
        if (l == 8) { P.init = INIT_DENSITY; P.init_off = D.c_w8row() + 256 * j; }
    }
    return P;
}

// ---------------------------------------------------------------------------------------------------------------
// packing: constant block + the two weight streams, each pass a run of zero-padded slot-major pairs
// ---------------------------------------------------------------------------------------------------------------
struct PackArgs {
    Dims D;
    const float *P;
    int dx, inputs;      // which program
    int n_pairs;
};

__global__ void layered_pack_consts(const Dims D, const float *__restrict__ P, float *__restrict__ out) {
    const int total = D.c_floats();
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (e < D.c_wout()) { if (e < D.F) v = P[D.w[8] + e]; }                               // fc_8.weight[0, :]
        else if (e < D.c_scal()) { const int c = (e - D.c_wout()) / D.Hp, k = (e - D.c_wout()) % D.Hp; if (k < D.H) v = P[D.w[10] + c * D.H + k]; }
        else if (e < D.c_scal() + 4) { const int c = e - D.c_scal(); v = c == 0 ? P[D.b[8]] : P[D.b[10] + c - 1]; }
        else if (e < D.c_bias(9) + D.Hp) {
            const int l = (e - D.c_bias(0)) / D.Fp, n = (e - D.c_bias(0)) % D.Fp;
            if (l < 9) { if (n < D.F) v = P[D.b[l] + (l == 8 ? 1 : 0) + n]; }
            else if (n < D.H) v = P[D.b[9] + n];
        }
        out[e] = v;
    }
}

__global__ void layered_pack_stream(const PackArgs a, float *__restrict__ out) {
    // one workgroup per 4-KiB slot (grid-stride; 16 slots per pair): find the pass that owns the pair (scalar walk,
    // <= ~50 passes) -- a narrow network has a dozen pairs, and one workgroup per pair left the chip idle for 65 us
    for (int unit = blockIdx.x; unit < a.n_pairs * 16; unit += gridDim.x) {
        const int pair = unit >> 4;
        int idx = 0, first = 0;
        Pass P = a.dx ? dx_pass(a.D, a.inputs, 0) : fwd_pass(a.D, 0);
        while (first + P.pairs() <= pair) {
            first += P.pairs(); ++idx;
            P = a.dx ? dx_pass(a.D, a.inputs, idx) : fwd_pass(a.D, idx);
        }
        const int kpc = 8 / P.nfb, total_kb = P.kb0 + P.kb1;
        float *dst = out + (int64_t)pair * (PAIR_BYTES / 4);
        for (int e = (unit & 15) * 1024 + threadIdx.x; e < (unit & 15) * 1024 + 1024; e += blockDim.x) {
            const int b = e * 4;                       // byte offset inside the pair
            const int chunk = b >> 15, slot = (b >> 12) & 7, in_slot = b & 4095;
            const int i = in_slot >> 7;                // row of the 32-row block
            const int c = ((in_slot & 127) >> 4) ^ ((i >> 1) & 7);    // logical k-group (chunk_slot_offset's swizzle)
            const int kk = 4 * c + ((in_slot & 15) >> 2);
            const int kbi = ((pair - first) * 2 + chunk) * kpc + slot / P.nfb, fb = slot % P.nfb;
            float v = 0.0f;
            if (kbi < total_kb && fb < P.blocks) {
                const int s = kbi < P.kb0 ? 0 : 1;
                const int k = 32 * (s ? kbi - P.kb0 : kbi) + kk;       // index inside the source plane
                const int n = 32 * (P.dst_fb0 + fb) + i;                 // index inside the destination plane
                const int L = s ? P.layer1 : P.layer0, c0 = s ? P.col01 : P.col00, cv = s ? P.cols_valid1 : P.cols_valid0;
                if (!P.transposed) {
                    if (n < P.rows_valid && k < cv) v = a.P[a.D.w[L] + (int64_t)(P.row0 + n) * a.D.in[L] + c0 + k];
                } else {
                    if (k < P.rows_valid && n < cv) v = a.P[a.D.w[L] + (int64_t)(P.row0 + k) * a.D.in[L] + c0 + n];
                }
            }
            dst[e] = v;
        }
    }
}

int stream_pairs(const Dims &D, int dx, int inputs) {
    int n = 0;
    const int np = dx ? dx_num_passes(D, inputs) : fwd_num_passes(D);
    for (int p = 0; p < np; ++p) n += (dx ? dx_pass(D, inputs, p) : fwd_pass(D, p)).pairs();
    return n;
}

// ---------------------------------------------------------------------------------------------------------------
// rows <-> planes
// ---------------------------------------------------------------------------------------------------------------
// plane[m][k] = k < E ? rows[m][k] : 0 for m < M, 0 for the padded rows (TF layout).  One workgroup per 32-sample tile
// (grid-stride), 128 columns at a time through LDS: the rows of a tile are one contiguous run of the input (read
// coalesced), the 16 slots of 128 columns one contiguous 16 KiB of the plane (written as 16-byte units in memory
// order, inverting tf_offset).  (The per-element version moved 2 TB/s.)
__global__ __launch_bounds__(256) void rows_to_plane_kernel(const float *__restrict__ rows, int64_t M, int64_t MP, int E, int W,
                                                            float *__restrict__ plane) {
    __shared__ float t[32 * 129];
    for (int64_t tile = blockIdx.x; tile < MP / 32; tile += gridDim.x) {
        for (int kc = 0; kc < W; kc += 128) {
            const int cw = E - kc < 128 ? (E - kc > 0 ? E - kc : 0) : 128;      // valid columns of this chunk
            __syncthreads();
            for (int idx = threadIdx.x; idx < 32 * cw; idx += 256) {
                const int r = idx / cw, c = idx - r * cw;
                const int64_t m = tile * 32 + r;
                t[r * 129 + c] = m < M ? rows[m * E + kc + c] : 0.0f;
            }
            __syncthreads();
            const int wc = W - kc < 128 ? W - kc : 128;                         // plane columns of this chunk
            for (int u = threadIdx.x; u < wc * 8; u += 256) {                   // 16-byte units: 64 per 256-float slot
                const int slot = kc / 8 + (u >> 6), w = u & 63;
                const int unit = w ^ (2 * (slot & 3));
                const int r = unit >> 1, k = 32 * (slot >> 2) + 8 * (slot & 3) + 4 * (unit & 1) - kc;
                f32x4 v;
                v.x = k + 0 < cw ? t[r * 129 + k + 0] : 0.0f;
                v.y = k + 1 < cw ? t[r * 129 + k + 1] : 0.0f;
                v.z = k + 2 < cw ? t[r * 129 + k + 2] : 0.0f;
                v.w = k + 3 < cw ? t[r * 129 + k + 3] : 0.0f;
                *reinterpret_cast<f32x4 *>(plane + tile * 32 * W + (int64_t)slot * 256 + 4 * w) = v;
            }
        }
    }
}
// plane[m][k] = PositionalEncoder(3, L, include_input).encode(x[m])[k] for k < E, 0 beyond / for the padded rows (TF layout):
// the encoder kernel and rows_to_plane_kernel in one -- RAW points in, input plane out; the (M, E) row-major encoding
// never exists.  One workgroup per 32-sample tile: the 32 x L x 3 (octave, channel) arguments get ONE sincosf each
// (bit-identical to posenc.hip's sinf / cosf per element: scripts/sincos_probe.hip, 0 of 2^24 arguments over 16 octaves
// differ), the tile's features are collected in LDS and leave as 16-byte units in memory order like rows_to_plane's.
// Layout of a row (positional_encoder.py:83-88): [x y z (if include_input) | sin(2^0 xyz) cos(2^0 xyz) | sin(2^1 xyz) ...].
constexpr int ENC_MAX_W = 256;
__global__ __launch_bounds__(256) void encode_to_plane_kernel(const float *__restrict__ x, int64_t M, int64_t MP, int L, int inc,
                                                              int E, int W, float *__restrict__ plane) {
    __shared__ float t[32 * (ENC_MAX_W + 1)];
    const int ld = W + 1, raw = inc ? 3 : 0;
    for (int64_t tile = blockIdx.x; tile < MP / 32; tile += gridDim.x) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 32 * (W - E); idx += 256) {       // zero tail [E, W)
            const int r = idx / (W - E), c = idx - r * (W - E);
            t[r * ld + E + c] = 0.0f;
        }
        for (int idx = threadIdx.x; idx < 32 * 3 * (L + (inc ? 1 : 0)); idx += 256) {
            const int r = idx % 32, j = idx / 32;          // j = 3 f' + c with f' = 0 the raw copy (if inc), then the octaves
            const int c = j % 3, fq = j / 3;
            const int64_t m = tile * 32 + r;
            const float v = m < M ? x[3 * m + c] : 0.0f;
            if (inc && fq == 0) { t[r * ld + c] = m < M ? v : 0.0f; continue; }
            const int f = fq - (inc ? 1 : 0);
            // sinf and cosf SEPARATELY, like posenc.hip (and posenc_backward): the raw entry is documented bit-identical to
            // PositionalEncoder.encode + the pre-encoded entry, and sincosf returning the same bits as the two calls is a
            // property of one ROCm device library, probed for 16 of the ~42 octaves this entry admits
            const float arg = ldexpf(v, f);                // 2^f * x, exact (:81, :87-88)
            const float sn = sinf(arg), cs = cosf(arg);
            t[r * ld + raw + 6 * f + c] = m < M ? sn : 0.0f;
            t[r * ld + raw + 6 * f + 3 + c] = m < M ? cs : 0.0f;
        }
        __syncthreads();
        for (int u = threadIdx.x; u < W * 8; u += 256) {                     // 16-byte units: 64 per 256-float slot
            const int slot = u >> 6, w = u & 63;
            const int unit = w ^ (2 * (slot & 3));
            const int r = unit >> 1, k = 32 * (slot >> 2) + 8 * (slot & 3) + 4 * (unit & 1);
            const f32x4 v = {t[r * ld + k], t[r * ld + k + 1], t[r * ld + k + 2], t[r * ld + k + 3]};
            *reinterpret_cast<f32x4 *>(plane + tile * 32 * W + (int64_t)slot * 256 + 4 * w) = v;
        }
    }
}
__global__ void plane_to_rows_kernel(const float *__restrict__ plane, int64_t M, int E, int W, float *__restrict__ rows) {
    const int64_t total = M * E;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / E;
        rows[e] = plane[tf_offset(W, m, (int)(e - m * E))];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the persistent kernel
// ---------------------------------------------------------------------------------------------------------------
struct WideArgs {
    Dims D;
    const char *stream;      // weight pairs of this program
    const float *consts;     // constant block
    float *rec, *grad;       // buffer 0 / buffer 1 (plane = buffer + offset * MP)
    int64_t M, MP;
    int n_passes, n_pairs, inputs;
    int record;                          // forward: a backward will read this record (ReLU bit planes are written)
    float *sigma, *rgb;                  // forward outputs
    const float *sigma_in, *rgb_in, *g_sigma, *g_rgb;   // reverse chain inputs
};

// 16 bytes of this lane from a 4-KiB block: wave-uniform base + the lane's group offset + q * 1024.  A PLAIN load: the
// first version issued these through inline asm (placement under control, no compiler waitcnt) -- and hipcc, which
// believes an asm output valid at once, copied the destinations around before the data arrived and re-used the
// registers of loads whose results a short pass never reads, which the late data then overwrote (memory faults
// for feat_dim 64; scripts/audit_asm_loads.py now checks asm-issued global loads as well).  The compiler's own
// waits for these loads sit at their first use, behind the pair's acquire, where nothing is outstanding any more.
__device__ __forceinline__ f32x4 load16_s(const char *base, unsigned voff, int imm) {
    return *reinterpret_cast<const f32x4 *>(base + voff + imm);
}

struct Ctx {
    const char *lds;
    int offq[4];
    unsigned voff[4];        // byte offset of this lane's 16-byte group in slot q of a 4-KiB block: ((2 i + h) << 4) ^ (32 q)
    int i, h;
    int64_t row0;            // first sample of this wavefront in the current tile (multiple of 32)
    int64_t m;               // this lane's sample
    bool valid;
};

// this wavefront's 32-sample tile of a plane: base + (row0 / 32) * (32 * width floats)
__device__ __forceinline__ const char *tile_of(const float *buf, int off, int width, int64_t MP, int64_t row0) {
    return reinterpret_cast<const char *>(buf + (int64_t)off * MP + row0 * width);
}

// ReLU decisions of NFB blocks of this lane -> NFB / 2 dwords (bit 16 (fb & 1) + r of word fb / 2 = register r of block fb
// is > 0, i.e. its bit pattern is not 0 after the ReLU): two vector instructions per value (mlp_device.h: save_mask)
template <int NFB>
__device__ __forceinline__ void pack_mask_words(const f32x16 *blk, unsigned (&w)[(NFB + 1) / 2]) {
#pragma unroll
    for (int k = 0; k < (NFB + 1) / 2; ++k) w[k] = 0u;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            unsigned t;
            asm("v_min_u32 %0, 1, %1" : "=v"(t) : "v"(blk[fb][r]));
            w[fb >> 1] |= t << (16 * (fb & 1) + r);
        }
}
// v where bit (fb, r) of the mask words is set, else +0 (mlp_backward.hip: masked)
__device__ __forceinline__ float keep_bit(unsigned word, int bit, float v) {
    int keep;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(word), "n"(bit));   // the bit, sign-extended to 0 / ~0 (`bit`: a constant after unrolling)
    return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & keep);
}
// this lane's mask words: plane + ((2 m + h) mw + w0) dwords
__device__ __forceinline__ unsigned *mask_words_of(float *rec, int mask_off, int64_t MP, int64_t m, int h, int mw, int w0) {
    return reinterpret_cast<unsigned *>(rec + (int64_t)mask_off * MP) + (2 * m + h) * mw + w0;
}

// B operands of one PAIR of the pass (2 * KPC k-blocks starting at k-block kb_first).  k-blocks past the pass's end
// re-read the last block: their weights are zero slots of the stream and every plane holds finite values.
template <int KPC>
__device__ __forceinline__ void load_b_pair(f32x16 (&b)[2 * KPC], const Pass &P, int kb_first, const char *src0,
                                            const char *src1, const Ctx &c) {
    const int total = P.kb0 + P.kb1;
#pragma unroll
    for (int j = 0; j < 2 * KPC; ++j) {
        int kb = kb_first + j;       // wave-uniform
        if (kb >= total) kb = total - 1;
        const char *base = kb < P.kb0 ? src0 + (size_t)kb * 4096 : src1 + (size_t)(kb - P.kb0) * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = load16_s(base, c.voff[q], q * 1024);
            b[j][4 * q + 0] = v.x; b[j][4 * q + 1] = v.y; b[j][4 * q + 2] = v.z; b[j][4 * q + 3] = v.w;
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // the loads stay here, a pair ahead of their use
}

// the same, one 16-byte load at a time: load n = 4 j + q of the pair (j = k-block of the pair, q = 16-byte group) -- issued
// from inside the MFMA groups of the previous pair (mma_slots' hook)
template <int KPC>
__device__ __forceinline__ void load_b_one(f32x16 (&b)[2 * KPC], const Pass &P, int kb_first, const char *src0,
                                           const char *src1, const Ctx &c, int n) {
    const int j = n >> 2, q = n & 3, total = P.kb0 + P.kb1;
    int kb = kb_first + j;       // wave-uniform
    if (kb >= total) kb = total - 1;
    const char *base = kb < P.kb0 ? src0 + (size_t)kb * 4096 : src1 + (size_t)(kb - P.kb0) * 4096;
    const f32x4 v = load16_s(base, c.voff[q], q * 1024);
    b[j][4 * q + 0] = v.x; b[j][4 * q + 1] = v.y; b[j][4 * q + 2] = v.z; b[j][4 * q + 3] = v.w;
}

// One pass over the current tile.  DX: the reverse-chain flavour (ReLU masks from the record, density-row init).
// pr0..pr2 carry the fc_out partial dot products across the passes of fc_9; `sig` the density pre-activation.
template <int NFB, bool DX>
__device__ __forceinline__ void run_pass(const Pass &P, const WideArgs &a, const Ctx &c, Pipe &pipe, float &pr0,
                                         float &pr1, float &pr2, float &sig, float dsig) {
    constexpr int KPC = 8 / NFB;
    const float *sbuf0 = P.src_buf0 ? a.grad : a.rec, *sbuf1 = P.src_buf1 ? a.grad : a.rec;
    const char *src0 = tile_of(sbuf0, P.src_off0, P.src_w0, a.MP, c.row0);
    const char *src1 = tile_of(sbuf1, P.src_off1, P.src_w1, a.MP, c.row0);
    const int pairs = P.pairs(), chunks = P.chunks();

    f32x16 bA[2 * KPC], bB[2 * KPC];
    const unsigned long long lt0 = LT_NOW();
    // the sources were stored by this very wavefront one pass ago (asm-issued stores): drain them, then read back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long lt1 = LT_NOW();
    unsigned long long lt_acq = 0;
    load_b_pair<KPC>(bA, P, 0, src0, src1, c);

    f32x16 acc[NFB];
    // initial accumulators: bias window / density row x dsigma' / zero.  ONE wave-uniform branch: a condition inside the
    // (fb, q) loops became a scalar branch per element group (the timeline of scripts/timeline_layered.py: 2 k cycles here
    // and 13 k in the epilogue of a 148 k-cycle pass, all of it s_cbranch).  Blocks the pass does not have start from the
    // first block's values: their weight slots are zero and nothing stores them.
    if (P.init == INIT_ZERO) {
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fb][r] = 0.0f;
    } else {
        const float *row = a.consts + P.init_off;
        const float s = (DX && P.init == INIT_DENSITY) ? dsig : 1.0f;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
            const float *rb = row + 32 * (fb < P.blocks ? fb : 0);     // scalar select
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(rb + 8 * q + 4 * c.h);
                acc[fb][4 * q + 0] = v.x * s; acc[fb][4 * q + 1] = v.y * s;
                acc[fb][4 * q + 2] = v.z * s; acc[fb][4 * q + 3] = v.w * s;
            }
        }
    }
    if (!DX && P.side == SIDE_DENSITY) sig = 0.0f;

    // DX: the ReLU decisions that mask this pass's result, NFB / 2 dwords from the record's bit planes (an unmasked stage
    // keeps everything: all ones, and the epilogue below has no branch).  Words past the plane's own (a ragged last
    // pass) re-read its last word: they mask blocks nothing stores.
    constexpr int NW = NFB / 2;
    unsigned mkb[NW];
    const bool masked = DX && P.mask_off >= 0;
    const int mwords = a.D.mw(), w0 = P.dst_fb0 / 2;
    const int nw_valid = mwords - w0 < NW ? mwords - w0 : NW;     // wave-uniform
    if (DX) {
        const unsigned *mp = masked ? mask_words_of(a.rec, P.mask_off, a.MP, c.m, c.h, mwords, w0) : nullptr;
#pragma unroll
        for (int k = 0; k < NW; ++k) mkb[k] = masked ? mp[k < nw_valid ? k : nw_valid - 1] : 0xffffffffu;
    }

    // the density row of fc_8 rides on the B operands streaming by (nerf.py:113-115): sigma' = W8[0, :] . h7 + b
    auto density = [&](const f32x16 (&b)[2 * KPC], int kb_first) {
        const float *w8 = a.consts + a.D.c_w8row();
#pragma unroll
        for (int j = 0; j < 2 * KPC; ++j) {
            const int kb = kb_first + j;
            if (kb < P.kb0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(w8 + 32 * kb + 8 * q + 4 * c.h);
                    sig = fmaf(w.x, b[j][4 * q + 0], sig); sig = fmaf(w.y, b[j][4 * q + 1], sig);
                    sig = fmaf(w.z, b[j][4 * q + 2], sig); sig = fmaf(w.w, b[j][4 * q + 3], sig);
                }
            }
        }
    };

    const unsigned long long lt2 = LT_NOW();
    for (int pr = 0; pr < pairs; pr += 2) {
        {   // even pair: operands bA, next pair's into bB
            const unsigned long long la = LT_NOW();
            const char *w = c.lds + pipe.acquire();
            lt_acq += LT_NOW() - la;
            // the next pair's operands are fetched from inside this pair's MFMA groups, one 16-byte load per 32 / (8 KPC)
            // groups -- in the 8-block passes (wide layers: many pairs; past the pass's end the eight loads re-read its
            // last block and are never multiplied: a branch per load, or a second flavour of the MFMA loop, costs more).
            // Narrow passes are one or two pairs long: they fetch in front of the MFMAs, and only what exists.
            auto next_b = [&](int g, int groups) {
                constexpr int NL = 8 * KPC;
                if (NFB == 8 && g % (groups / NL) == 0) load_b_one<KPC>(bB, P, (pr + 1) * 2 * KPC, src0, src1, c, g / (groups / NL));
            };
            if (NFB < 8 && pr + 1 < pairs) load_b_pair<KPC>(bB, P, (pr + 1) * 2 * KPC, src0, src1, c);
            if (!DX && P.side == SIDE_DENSITY) density(bA, pr * 2 * KPC);
            mma_slots<NFB, KPC, 0, 16, false>(acc, bA, w, c.offq, &pipe, next_b);
            if (2 * pr + 1 < chunks) mma_slots<NFB, KPC>(acc, bA + KPC, w + CHUNK_BYTES, c.offq);
            pipe.issue_done();
        }
        if (pr + 1 < pairs) {   // odd pair: operands bB, next pair's into bA
            const unsigned long long la = LT_NOW();
            const char *w = c.lds + pipe.acquire();
            lt_acq += LT_NOW() - la;
            auto next_b = [&](int g, int groups) {
                constexpr int NL = 8 * KPC;
                if (NFB == 8 && g % (groups / NL) == 0) load_b_one<KPC>(bA, P, (pr + 2) * 2 * KPC, src0, src1, c, g / (groups / NL));
            };
            if (NFB < 8 && pr + 2 < pairs) load_b_pair<KPC>(bA, P, (pr + 2) * 2 * KPC, src0, src1, c);
            if (!DX && P.side == SIDE_DENSITY) density(bB, (pr + 1) * 2 * KPC);
            mma_slots<NFB, KPC, 0, 16, false>(acc, bB, w, c.offq, &pipe, next_b);
            if (2 * pr + 3 < chunks) mma_slots<NFB, KPC>(acc, bB + KPC, w + CHUNK_BYTES, c.offq);
            pipe.issue_done();
        }
    }

    // ---- epilogue
    const unsigned long long lt3 = LT_NOW();
    LT_ADD(0, lt1 - lt0); LT_ADD(1, lt2 - lt1); LT_ADD(2, lt_acq); LT_ADD(3, lt3 - lt2 - lt_acq); LT_ADD(6, 1ull);
    // branch-free: ReLU as max(v, floor) with floor = 0 | -inf in a scalar register; the reverse chain always selects
    const float floor = P.relu ? 0.0f : -__builtin_inff();
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[fb][r];
            if (!DX) asm("v_max_f32 %0, %1, %2" : "=v"(v) : "s"(floor), "v"(acc[fb][r]));
            else v = keep_bit(mkb[fb >> 1], 16 * (fb & 1) + r, v);
            acc[fb][r] = v;
        }
    if (!DX && a.record && P.mask_off >= 0) {   // the ReLU decisions of this pass, for the reverse chain (bit planes)
        unsigned w[NW];
        pack_mask_words<NFB>(acc, w);
        unsigned *mp = mask_words_of(a.rec, P.mask_off, a.MP, c.m, c.h, mwords, w0);
#pragma unroll
        for (int k = 0; k < NW; ++k)
            if (k < nw_valid) mp[k] = w[k];
        // (h9 is half as wide as the planes are strided for: its unused words are written too -- every byte of a record is
        // a function of the inputs, tests/test_gpu_determinism.py compares them all)
        if (P.last)
            for (int k = w0 + nw_valid; k < mwords; ++k) mp[k - w0] = 0u;
    }
    float *dbuf = P.dst_buf ? a.grad : a.rec;
    float *dplane = dbuf + (int64_t)P.dst_off * a.MP + (int64_t)P.dst_fb0 * 1024;
    // (block counts are wave-uniform: whole store groups are skipped for blocks the plane does not have)
    if (P.blocks >= NFB) save_plane<NFB, true>(dplane, P.dst_w, c.m, c.h, acc);
    else {
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
            if (fb < P.blocks) save_plane<1, true>(dplane + fb * 1024, P.dst_w, c.m, c.h, acc + fb);
    }
    if (!DX && P.side == SIDE_DENSITY) {
        float s = sig + __shfl_xor(sig, 32, WAVE);
        s = fmaxf(s + a.consts[a.D.c_scal()], 0.0f);           // relu(x[:, 0]) (nerf.py:115)
        if (c.valid && c.h == 0) a.sigma[c.m] = s;
    }
    if (!DX && P.side == SIDE_FCOUT) {   // rgb = sigmoid(fc_out(h9)) (nerf.py:119), partial dots over the passes of fc_9
        // (three scalars: an array handed through the three instantiations of this function ends up in scratch)
        if (P.first) pr0 = pr1 = pr2 = 0.0f;
        const float *w = a.consts + a.D.c_wout() + 32 * P.dst_fb0;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
            if (fb < P.blocks) {
                pr0 += block_dot(w + 32 * fb, acc[fb], c.h);
                pr1 += block_dot(w + a.D.Hp + 32 * fb, acc[fb], c.h);
                pr2 += block_dot(w + 2 * a.D.Hp + 32 * fb, acc[fb], c.h);
            }
        if (P.last) {
            const float *bo = a.consts + a.D.c_scal() + 1;
            const float y0 = 1.0f / (1.0f + expf(-(pr0 + __shfl_xor(pr0, 32, WAVE) + bo[0])));
            const float y1 = 1.0f / (1.0f + expf(-(pr1 + __shfl_xor(pr1, 32, WAVE) + bo[1])));
            const float y2 = 1.0f / (1.0f + expf(-(pr2 + __shfl_xor(pr2, 32, WAVE) + bo[2])));
            if (c.valid && c.h == 0) { a.rgb[3 * c.m] = y0; a.rgb[3 * c.m + 1] = y1; a.rgb[3 * c.m + 2] = y2; }
        }
    }
    LT_ADD(4, LT_NOW() - lt3);
}

template <bool DX>
__global__ __launch_bounds__(256, 1) void layered_kernel(const WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Ctx c;
    c.lds = lds;
    c.i = lane & 31; c.h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        c.offq[q] = chunk_slot_offset(c.i, 2 * q + c.h);
        c.voff[q] = ((unsigned)(2 * c.i + c.h) << 4) ^ (32u * q);
    }
    Pipe pipe;
    pipe.src_wave = a.stream + wave * 8192;
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 8192u;
    pipe.issued = 0; pipe.issue_pos = 0; pipe.consumed = 0;
    pipe.n_pairs = a.n_pairs; pipe.skip_mask = 0;
    // the pass program, tabulated behind the ring: thread p evaluates pass p once
    int *ptab = reinterpret_cast<int *>(lds + RING_SLOTS * CHUNK_BYTES);
    // (both programs.  While the reverse chain still fetched float masks -- 32 loads in front of a pass's last pair -- it
    // came out 1-2.5 % slower with the table: its scalar program had overlapped their latency.  With bit planes: +0.5 %)
    const bool tabulated = a.n_passes <= PASS_TABLE_MAX;
    if (tabulated && tid < a.n_passes) {
        const Pass P = DX ? dx_pass(a.D, a.inputs, tid) : fwd_pass(a.D, tid);
        int k = 0;
#define X(f) ptab[tid * PASS_ROW + k++] = P.f;
        PASS_FIELDS(X)
#undef X
    }
    __syncthreads();
    pipe.issue();

    const int64_t ntiles = a.MP / TILE_SAMPLES;
#ifdef X_STAGGER_TICKS   // experiment: workgroups start spread over X_STAGGER_TICKS x 10 ns so that their epilogues do not coincide
    {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)blockIdx.x * X_STAGGER_TICKS / gridDim.x;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        c.row0 = tile * TILE_SAMPLES + wave * 32;
        c.m = c.row0 + c.i;
        c.valid = c.m < a.M;
        float pr0 = 0.0f, pr1 = 0.0f, pr2 = 0.0f, sig = 0.0f, dsig = 0.0f;
        if (DX) {
            // ---- heads (nerf.py:115, :119): d y10 = g_rgb rgb (1 - rgb), dsigma' = g_sigma . [sigma > 0]; then
            // dY9 = (W_out^T d y10) . [h9 > 0] on the vector ALU, block by block
            const int64_t mc = c.valid ? c.m : a.M - 1;
            float gy[3], yv[3], gv[3];      // (all loads first, from the clamped row: no load-then-wait chains under exec masks)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { yv[ch] = a.rgb_in[3 * mc + ch]; gv[ch] = a.g_rgb[3 * mc + ch]; }
            const float sg = a.sigma_in[mc], gsg = a.g_sigma[mc];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float t = gv[ch] * yv[ch] * (1.0f - yv[ch]);
                gy[ch] = c.valid ? t : 0.0f;
            }
            dsig = (c.valid && sg > 0.0f) ? gsg : 0.0f;
            if (c.h == 0) {
                const f32x4 g4 = {gy[0], gy[1], gy[2], 0.0f};
                *reinterpret_cast<f32x4 *>(a.grad + (int64_t)a.D.g_gy() * a.MP + 4 * c.m) = g4;
                a.grad[(int64_t)a.D.g_dsig() * a.MP + c.m] = dsig;
            }
            const float *wout = a.consts + a.D.c_wout();
            const unsigned *m9 = mask_words_of(a.rec, a.D.r_mask(8), a.MP, c.m, c.h, a.D.mw(), 0);
            float *d9 = a.grad + (int64_t)a.D.g_dy9() * a.MP;
            for (int fb = 0; fb < a.D.Hp / 32; ++fb) {
                const unsigned bits9 = (m9[fb >> 1] >> (16 * (fb & 1))) & 0xffffu;     // [h9 > 0] of this block's 16 registers
                f32x16 x;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k0 = 32 * fb + 8 * q + 4 * c.h;
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wout + k0);
                    const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wout + a.D.Hp + k0);
                    const f32x4 w2 = *reinterpret_cast<const f32x4 *>(wout + 2 * a.D.Hp + k0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = fmaf(w2[j], gy[2], fmaf(w1[j], gy[1], w0[j] * gy[0]));
                        x[4 * q + j] = ((bits9 >> (4 * q + j)) & 1u) ? v : 0.0f;
                    }
                }
                save_plane<1, true>(d9 + fb * 1024, a.D.Hp, c.m, c.h, &x);
            }
        }
        for (int p = 0; p < a.n_passes; ++p) {
            const unsigned long long lp = LT_NOW();
            Pass P;
            if (tabulated) {
                const int v = ptab[p * PASS_ROW + (lane & 31)];
                int k = 0;
#define X(f) P.f = __builtin_amdgcn_readlane(v, k++);
                PASS_FIELDS(X)
#undef X
            } else {
                P = DX ? dx_pass(a.D, a.inputs, p) : fwd_pass(a.D, p);
            }
            // (the stamp must follow the program: a scalar of it orders the reads)
            LT_ADD(5, LT_NOW() - lp + (unsigned long long)(P.nfb & 0));
            if (P.nfb == 8) run_pass<8, DX>(P, a, c, pipe, pr0, pr1, pr2, sig, dsig);
            else if (P.nfb == 4) run_pass<4, DX>(P, a, c, pipe, pr0, pr1, pr2, sig, dsig);
            else run_pass<2, DX>(P, a, c, pipe, pr0, pr1, pr2, sig, dsig);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// networks whose activations fit the register file: feat_dim 97..128 (NSB = 2) and 225..256 (NSB = 1), view_dir_dim
// <= 32, pos_dim <= 96 -- e.g. NeRF(63, 27, 128), or NeRF(75..99, 27, 256) = coord_encode_level 12..15, which the fused
// family (pos_dim <= 64) does not take
// ---------------------------------------------------------------------------------------------------------------
// NSB sample blocks x NFB = 8 / NSB feature blocks per wavefront: 128 accumulator + 128 activation registers either way.
// The D fragment of a layer is the B fragment of the next again, nothing is parked -- planes are written only for a
// backward that will read them (RECORD) -- and with NSB = 2 every A fragment feeds two sample blocks (mma_slots2: half
// the LDS reads and half the weight stream per MFMA).  Same streams, same planes, same dW path as the general kernel
// above, whose pass programs these kernels walk with everything unrolled (PB = position blocks, 1..3):
//   forward   fc_in | fc_1..4 | fc_5 (pos blocks first) | fc_6..8 | fc_9 (fc_8's output blocks, then the direction)
//   reverse   fc_9^T | [g_view_dir] | fc_8^T .. fc_1^T | [g_pos: dY0 blocks, then dY5 blocks re-read from their plane]
constexpr int REG_LDS = RING_SLOTS * CHUNK_BYTES + 16384;   // ring + the constant block (<= 4096 floats)

__host__ __device__ inline bool reg_ok(const Dims &D) {
    // (four position blocks -- coord_encode_level 16: pos_dim 99 -- where the encoded position is one sample block or
    // the network is narrow; with two sample blocks of 128 features its 128 registers do not fit)
    // (two direction blocks -- dir_encode_level 5..10: view_dir_dim 33..63 -- for the 256-feature forward only: its
    // reverse chain is the general one, and the narrow networks are not reachable from the reference's configs)
    return ((D.Fp == 64 && D.Hp == 32) || (D.Fp == 128 && D.Hp == 64) || (D.Fp == 256 && D.Hp == 128)) &&
           (D.Dp == 32 || (D.Dp == 64 && D.Fp == 256)) &&
           D.Pp <= (D.Fp == 128 ? 96 : 128) && D.c_floats() <= 4096;
}

template <int N, class F> __device__ __forceinline__ void static_for(F f) {     // f(integral_constant<0>) .. f(<N-1>)
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// this lane's 16 registers of block `blk` of a 32-sample tile (the lane's own slots, as save_plane stores them)
__device__ __forceinline__ f32x16 load_block(const float *tile, int blk, int i, int h) {
    f32x16 x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + (blk * 4 + q) * 256 + 4 * ((2 * i + h) ^ (2 * q)));
        x[4 * q + 0] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
    return x;
}

template <int NSB, int NFB, int NB>
__device__ __forceinline__ void bias_init(f32x16 (*acc)[NFB], const float *bias, int h) {
#pragma unroll
    for (int fb = 0; fb < NB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(bias + 32 * fb + 8 * q + 4 * h);
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                acc[sb][fb][4 * q + 0] = v.x; acc[sb][fb][4 * q + 1] = v.y; acc[sb][fb][4 * q + 2] = v.z; acc[sb][fb][4 * q + 3] = v.w;
            }
        }
}

// The k-blocks seq(sb, 0 .. N-1) of one pass against NFBC accumulator blocks: chunks of KPCX k-blocks (slot stride
// STRIDE), two chunks per pair.  `w0` = the already acquired first pair, or null.  `before(c)` runs in front of chunk c
// (a hook: the reverse chain re-reads dY5 between the dY0 and the dY5 blocks of g_pos).
// FRESH: the accumulators start from zero (C = 0 in the first MFMAs).
// STORES > 0 (one sample block): 2 STORES plane stores of `st` ride between the MFMA groups of the first chunks of the first two
// pairs (PlaneStore: the blocks stored are the B operands of this very run and stay untouched through it).
template <int NSB, int NFBC, int STRIDE, int KPCX, int N, bool FRESH, bool ACQ_FIRST, int STORES = 0, class Seq, class Before>
__device__ __forceinline__ void run_blocks(f32x16 *acc0, f32x16 *acc1, Seq seq, Before before, const char *w0, const char *lds,
                                           Pipe &pipe, const int (&offq)[4], const PlaneStore *st = nullptr) {
    constexpr int CH = (N + KPCX - 1) / KPCX, LAST = N - (CH - 1) * KPCX;   // chunks; k-blocks of the last one
    static_assert(STORES == 0 || (NSB == 1 && CH >= 3), "spread stores: one sample block, two pairs to carry them");
    const char *w = w0;       // (ACQ_FIRST: the first pair is acquired here; a template flag, not `w0 == nullptr`: no run-time
                              // branch may sit around an acquire, and LDS address 0 is a valid pointer)
    static_for<CH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (c % 2 == 0 && (c > 0 || ACQ_FIRST)) w = lds + pipe.acquire();
        before(c);
        auto b = [&](int sb, int kb) -> const f32x16 & { return seq(sb, c * KPCX + kb); };
        const char *wc = w + (c % 2) * CHUNK_BYTES;
        constexpr int NKBX = c == CH - 1 ? LAST : KPCX;
        if constexpr (c % 2 == 0 && STORES > 0 && c < 4)
            mma_slots2<NFBC, NKBX, STRIDE, 0, 16, (FRESH && c == 0), NSB, STORES, 2 * STORES, 2 * STORES>(acc0, acc1, b, wc, offq, &pipe, st,
                                                                                                     (c / 2) * STORES);
        else if constexpr (c % 2 == 0) mma_slots2<NFBC, NKBX, STRIDE, 0, 16, (FRESH && c == 0), NSB>(acc0, acc1, b, wc, offq, &pipe);
        else mma_slots2<NFBC, NKBX, STRIDE, 0, 0, false, NSB>(acc0, acc1, b, wc, offq);
        if (c % 2 == 1 || c == CH - 1) pipe.issue_done();
    });
}

// (NSB sample blocks, NFB feature blocks) per wavefront: (1, 8) = feat_dim 225..256, (2, 4) = 97..128, (2, 2) = 33..64
// DB = 32-wide blocks of the encoded direction (1: view_dir_dim <= 32; 2: <= 64, instantiated for NSB = 1 only)
template <int NSB, int NFB, int PB, bool RECORD, int DB = 1>
__global__ __launch_bounds__(256, 1) void reg_forward_kernel(const WideArgs a) {
    // fc_9's pass is packed with at least two accumulator blocks per k-block (pass_nfb)
    constexpr int HB = NFB / 2, KPC = 8 / NFB, S9 = HB < 2 ? 2 : HB, KPC9 = 8 / S9, FP = 32 * NFB, HP = 32 * HB, TILE = 128 * NSB;
    static_assert(NSB * NFB <= 8 && NFB >= 2, "register budget");
    // The stores of a recorded plane stay at the layer seam.  Spread between the MFMA groups of the layer that multiplies the
    // stored blocks (PlaneStore, as in the fused family's record forward and in reg_dx_kernel below) they cost the
    // one-sample-block kernels 1.2 % instead of gaining: A/B on one box, NeRF(75,27,256) at 786 432 samples, record forward
    // 7.20 / 7.24 ms spread vs 7.12 / 7.13 ms at the seam (NeRF(99,27,256): 7.48 vs 7.35) -- round 5, after the trunk layers
    // were peeled (before that the spread variants also spilled 160..250 B); two sample blocks: 2.13 vs 2.10 ms for feat 128.
    // -DX_REG_SPREAD_STORES rebuilds the spread variant.
#ifdef X_REG_SPREAD_STORES
    constexpr bool SPREAD = RECORD && NSB == 1 && NFB == 8;
#else
    constexpr bool SPREAD = false;
#endif
    constexpr int SPREAD_STORES = SPREAD ? 2 * NFB : 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float *cb = reinterpret_cast<float *>(lds + RING_SLOTS * CHUNK_BYTES);
    for (int e = tid; e < a.D.c_floats() / 4; e += 256)
        reinterpret_cast<f32x4 *>(cb)[e] = reinterpret_cast<const f32x4 *>(a.consts)[e];
    int offq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) offq[q] = chunk_slot_offset(i, 2 * q + h);
    Pipe pipe;
    pipe.src_wave = a.stream + wave * 8192;
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 8192u;
    pipe.issued = 0; pipe.issue_pos = 0; pipe.consumed = 0;
    pipe.n_pairs = a.n_pairs; pipe.skip_mask = 0;
    __syncthreads();
    pipe.issue();
    const Dims &D = a.D;
    const int64_t MP = a.MP;
    auto plane = [&](int off) { return a.rec + (int64_t)off * MP; };
    auto none = [](int) {};

    for (int64_t tile = blockIdx.x; tile < MP / TILE; tile += gridDim.x) {
        const int64_t row0 = tile * TILE + wave * (32 * NSB);
        int64_t m[NSB];
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) m[sb] = row0 + 32 * sb + i;
        f32x16 pe[NSB][PB];
        auto load_pe = [&]() {
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
                for (int b = 0; b < PB; ++b) pe[sb][b] = load_block(plane(D.r_pe()) + (row0 + 32 * sb) * D.Pp, b, i, h);
        };
        load_pe();
        f32x16 acc[NSB][NFB], act[NSB][NFB];

        // ---- fc_in (nerf.py:102)
        {
            const char *w = lds + pipe.acquire();
            bias_init<NSB, NFB, NFB>(acc, cb + D.c_bias(0), h);
            auto P = [&](int sb, int kb) -> const f32x16 & { return pe[sb][kb]; };
            run_blocks<NSB, NFB, NFB, KPC, PB, false, false>(acc[0], acc[NSB - 1], P, none, w, lds, pipe, offq);
        }
        // ---- fc_1 .. fc_8 (:103-113); skip connection at fc_5, pos FIRST (:108)
        float sig[NSB];
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) sig[sb] = 0.0f;
        // one trunk layer l = 1..8: ReLU of the previous layer's accumulators, record, bias, multiply.  SKIP (l == 5: the encoded
        // position first) and DENSITY (l == 8) are compile-time: as run-time branches inside one loop body they cost the
        // one-sample-block record kernels their register allocation once the stores ride between the MFMAs
        auto trunk_layer = [&](int l, auto skip_tag, auto density_tag) {
            constexpr bool SKIP = decltype(skip_tag)::value, DENSITY = decltype(density_tag)::value;
            const char *w = lds + pipe.acquire();
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) act[sb][fb][r] = relu1(acc[sb][fb][r]);
            PlaneStore st;
            if (SPREAD) st.open(plane(D.r_h(l - 1)), FP, m[0], h, act[0]);
            if (RECORD) {
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) {
                    if (!SPREAD) save_plane<NFB, true>(plane(D.r_h(l - 1)), FP, m[sb], h, act[sb]);
                    unsigned w[NFB / 2];          // [h(l-1) > 0] as bits, for the reverse chain
                    pack_mask_words<NFB>(act[sb], w);
                    unsigned *mp = mask_words_of(a.rec, D.r_mask(l - 1), MP, m[sb], h, NFB / 2, 0);
#pragma unroll
                    for (int k = 0; k < NFB / 2; ++k) mp[k] = w[k];
                }
            }
            if (DENSITY) {   // density row of fc_8
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) sig[sb] = half_dot<NFB>(cb + D.c_w8row(), act[sb], h);
            }
            bias_init<NSB, NFB, NFB>(acc, cb + D.c_bias(l), h);
            if constexpr (SKIP) {
                // (the encoded position comes back from its plane -- L2-hot -- instead of living in 16 NSB PB registers
                // through fc_1..fc_4)
                load_pe();
                auto S5 = [&](int sb, int k) -> const f32x16 & { return k < PB ? pe[sb][k < PB ? k : 0] : act[sb][k >= PB ? k - PB : 0]; };
                run_blocks<NSB, NFB, NFB, KPC, PB + NFB, false, false, SPREAD_STORES>(acc[0], acc[NSB - 1], S5, none, w, lds, pipe, offq, &st);
            } else {
                auto A = [&](int sb, int k) -> const f32x16 & { return act[sb][k]; };
                run_blocks<NSB, NFB, NFB, KPC, NFB, false, false, SPREAD_STORES>(acc[0], acc[NSB - 1], A, none, w, lds, pipe, offq, &st);
            }
        };
        for (int l = 1; l <= 4; ++l) trunk_layer(l, std::false_type(), std::false_type());
        trunk_layer(5, std::true_type(), std::false_type());
        for (int l = 6; l <= 7; ++l) trunk_layer(l, std::false_type(), std::false_type());
        trunk_layer(8, std::false_type(), std::true_type());
        // ---- fc_9 on cat([x[:, 1:], view_dir]) (:116-118); fc_8 has no ReLU (:113)
        f32x16 a9[NSB][HB];
        {
            const char *w = lds + pipe.acquire();
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) act[sb][fb] = acc[sb][fb];
            PlaneStore st;
            if (SPREAD) st.open(plane(D.r_h(8)), FP, m[0], h, act[0]);
            if (RECORD && !SPREAD) {
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) save_plane<NFB, true>(plane(D.r_h(8)), FP, m[sb], h, act[sb]);
            }
            f32x16 de[NSB][DB];   // (fetched here, L2-hot, instead of living in 16 NSB DB registers through fc_in .. fc_8)
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
                for (int b = 0; b < DB; ++b) de[sb][b] = load_block(plane(D.r_de()) + (row0 + 32 * sb) * (32 * DB), b, i, h);
#pragma unroll
            for (int fb = 0; fb < HB; ++fb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(cb + D.c_bias(9) + 32 * fb + 8 * q + 4 * h);
#pragma unroll
                    for (int sb = 0; sb < NSB; ++sb) {
                        a9[sb][fb][4 * q + 0] = v.x; a9[sb][fb][4 * q + 1] = v.y; a9[sb][fb][4 * q + 2] = v.z; a9[sb][fb][4 * q + 3] = v.w;
                    }
                }
            auto Cat9 = [&](int sb, int k) -> const f32x16 & { return k < NFB ? act[sb][k < NFB ? k : 0] : de[sb][k >= NFB ? k - NFB : 0]; };
            run_blocks<NSB, HB, S9, KPC9, NFB + DB, false, false, SPREAD_STORES>(a9[0], a9[NSB - 1], Cat9, none, w, lds, pipe, offq, &st);
        }
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
            for (int fb = 0; fb < HB; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) a9[sb][fb][r] = relu1(a9[sb][fb][r]);
            if (RECORD) {
                save_plane<HB, true>(plane(D.r_h9()), HP, m[sb], h, a9[sb]);
                unsigned w[(HB + 1) / 2];
                pack_mask_words<HB>(a9[sb], w);
                unsigned *mp = mask_words_of(a.rec, D.r_mask(8), MP, m[sb], h, NFB / 2, 0);
#pragma unroll
                for (int k = 0; k < NFB / 2; ++k) mp[k] = k < (HB + 1) / 2 ? w[k < (HB + 1) / 2 ? k : 0] : 0u;   // (unused words: zeros)
            }
            float s = sig[sb] + __shfl_xor(sig[sb], 32, WAVE);
            s = fmaxf(s + cb[D.c_scal()], 0.0f);                      // relu(x[:, 0]) (:115)
            float y[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {                           // sigmoid(fc_out(h9)) (:119)
                float p = half_dot<HB>(cb + D.c_wout() + ch * HP, a9[sb], h);
                p += __shfl_xor(p, 32, WAVE);
                y[ch] = 1.0f / (1.0f + expf(-(p + cb[D.c_scal() + 1 + ch])));
            }
            if (m[sb] < a.M && h == 0) {
                a.sigma[m[sb]] = s;
                a.rgb[3 * m[sb] + 0] = y[0]; a.rgb[3 * m[sb] + 1] = y[1]; a.rgb[3 * m[sb] + 2] = y[2];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The reverse chain of the narrow networks (NSB = 2): written out for its one geometry -- 128 accumulator + 128
// activation + 128 ReLU-mask registers leave hipcc no slack, and every generic formulation tried (the run_blocks helper
// above, run-time FRESH, masks loaded at the seam, ...) spilled 20-150 registers and ran 5-15 % slower.  Networks of
// 225..256 features take the general reverse chain (layered_kernel<true>: 0.79-0.80 of peak) over the planes the
// register-resident forward recorded.
template <int NFB, int PB, bool IG>
__global__ __launch_bounds__(256, 1) void narrow_dx_kernel(const WideArgs a) {
    constexpr int HB = NFB / 2, FP = 32 * NFB, HP = 32 * HB;   // NFB = 4: feat_dim 97..128; NFB = 2: 33..64
    static_assert(NFB == 4 || NFB == 2, "two sample blocks x NFB feature blocks");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float *cb = reinterpret_cast<float *>(lds + RING_SLOTS * CHUNK_BYTES);
    for (int e = tid; e < a.D.c_floats() / 4; e += 256)
        reinterpret_cast<f32x4 *>(cb)[e] = reinterpret_cast<const f32x4 *>(a.consts)[e];
    int offq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) offq[q] = chunk_slot_offset(i, 2 * q + h);
    Pipe pipe;
    pipe.src_wave = a.stream + wave * 8192;
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 8192u;
    pipe.issued = 0; pipe.issue_pos = 0; pipe.consumed = 0;
    pipe.n_pairs = a.n_pairs; pipe.skip_mask = 0;
    __syncthreads();
    pipe.issue();
    const Dims &D = a.D;
    const int64_t MP = a.MP;
    auto gplane = [&](int off) { return a.grad + (int64_t)off * MP; };

    for (int64_t tile = blockIdx.x; tile < MP / 256; tile += gridDim.x) {
        const int64_t row0 = tile * 256 + wave * 64;
        const int64_t m[2] = {row0 + i, row0 + 32 + i};
        // ---- heads (nerf.py:115, :119) and dY9 = (W_out^T d y10) . [h9 > 0]
        float dsig[2];
        f32x16 d9[2][HB];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const bool valid = m[sb] < a.M;
            const int64_t mc = valid ? m[sb] : a.M - 1;
            float gy[3], yv[3], gv[3];      // (all loads first, from the clamped row: no load-then-wait chains under exec masks)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { yv[ch] = a.rgb_in[3 * mc + ch]; gv[ch] = a.g_rgb[3 * mc + ch]; }
            const float sg = a.sigma_in[mc], gsg = a.g_sigma[mc];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float t = gv[ch] * yv[ch] * (1.0f - yv[ch]);
                gy[ch] = valid ? t : 0.0f;
            }
            dsig[sb] = (valid && sg > 0.0f) ? gsg : 0.0f;
            if (h == 0) {
                const f32x4 g4 = {gy[0], gy[1], gy[2], 0.0f};
                *reinterpret_cast<f32x4 *>(gplane(D.g_gy()) + 4 * m[sb]) = g4;
                gplane(D.g_dsig())[m[sb]] = dsig[sb];
            }
            unsigned b9[(HB + 1) / 2];      // [h9 > 0]
#pragma unroll
            for (int k = 0; k < (HB + 1) / 2; ++k) b9[k] = mask_words_of(a.rec, D.r_mask(8), MP, m[sb], h, NFB / 2, 0)[k];
#pragma unroll
            for (int fb = 0; fb < HB; ++fb) {
                f32x16 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k0 = 32 * fb + 8 * q + 4 * h;
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(cb + D.c_wout() + k0);
                    const f32x4 w1 = *reinterpret_cast<const f32x4 *>(cb + D.c_wout() + HP + k0);
                    const f32x4 w2 = *reinterpret_cast<const f32x4 *>(cb + D.c_wout() + 2 * HP + k0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[4 * q + j] = fmaf(w2[j], gy[2], fmaf(w1[j], gy[1], w0[j] * gy[0]));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) d9[sb][fb][r] = keep_bit(b9[fb >> 1], 16 * (fb & 1) + r, v[r]);
            }
            save_plane<HB, true>(gplane(D.g_dy9()), HP, m[sb], h, d9[sb]);
        }
        f32x16 acc[2][NFB], act[2][NFB];
        unsigned mk[2][NFB / 2];        // the ReLU decisions of the next seam, from the record's bit planes
        auto D9 = [&](int sb, int kb) -> const f32x16 & { return d9[sb][kb]; };
        // ---- d y8[1:] = W9[:, :F]^T dY9
        {
            const char *w = lds + pipe.acquire();
            mma_slots2<NFB, HB, NFB, 0, 16, true>(acc[0], acc[1], D9, w, offq, &pipe);
            pipe.issue_done();
        }
        if (IG) {   // g_view_dir = W9[:, F:]^T dY9 (nerf.py:116)
            const char *w = lds + pipe.acquire();
            f32x16 gd[2];
            mma_slots2<1, HB, 2, 0, 16, true>(&gd[0], &gd[1], D9, w, offq, &pipe);
            pipe.issue_done();
            save_plane<1, true>(gplane(D.g_gd()), 32, m[0], h, &gd[0]);
            save_plane<1, true>(gplane(D.g_gd()), 32, m[1], h, &gd[1]);
        }
        // ReLU masks, fetched one stage ahead of the seam that applies them
        auto load_masks = [&](int l) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int k = 0; k < NFB / 2; ++k) mk[sb][k] = mask_words_of(a.rec, D.r_mask(l), MP, m[sb], h, NFB / 2, 0)[k];
        };
        auto apply_masks = [&](int sb) {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) act[sb][fb][r] = keep_bit(mk[sb][fb >> 1], 16 * (fb & 1) + r, acc[sb][fb][r]);
        };
        load_masks(7);
        // ---- l = 8 .. 1: dY(l-1) = (W_l^T dY(l)) . [h(l-1) > 0]; the accumulators entering stage l hold dY(l) unmasked
        for (int l = 8; l >= 1; --l) {
            const char *w = lds + pipe.acquire();
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                if (l == 8) {
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb) act[sb][fb] = acc[sb][fb];
                } else apply_masks(sb);
            }
            if (l < 8) load_masks(l - 1);   // masks of the next seam: h(l-1)
            PlaneStore st[2];     // dY(l) leaves between the MFMA groups of the first chunk
            st[0].open(gplane(D.g_dy(l)), FP, m[0], h, act[0]);
            st[1].open(gplane(D.g_dy(l)), FP, m[1], h, act[1]);
            auto A0 = [&](int sb, int kb) -> const f32x16 & { return act[sb][kb]; };
            auto A2 = [&](int sb, int kb) -> const f32x16 & { return act[sb][(2 + kb) % NFB]; };
            if (l == 8) {   // + W8[0, :] dsigma' (the density row)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 wv = *reinterpret_cast<const f32x4 *>(cb + D.c_w8row() + 32 * fb + 8 * q + 4 * h);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[sb][fb][4 * q + j] = wv[j] * dsig[sb];
                        }
                mma_slots2<NFB, 2, NFB, 0, 16, false, 2, 8 * NFB, 4 * NFB, 8 * NFB>(acc[0], acc[1], A0, w, offq, &pipe, st, 0);
            } else {
                mma_slots2<NFB, 2, NFB, 0, 16, true, 2, 8 * NFB, 4 * NFB, 8 * NFB>(acc[0], acc[1], A0, w, offq, &pipe, st, 0);
            }
            // (NFB = 2: the layer's two k-blocks are the first half of the pair's first chunk)
            if constexpr (NFB == 4) mma_slots2<4, 2, 4>(acc[0], acc[1], A2, w + CHUNK_BYTES, offq);
            pipe.issue_done();
        }
        // ---- dY0
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            apply_masks(sb);
            save_plane<NFB, true>(gplane(D.g_dy(0)), FP, m[sb], h, act[sb]);
        }
        if (IG) {   // g_pos = W_in^T dY0 + W5[:, :E_p]^T dY5 (nerf.py:102, :108); dY5 back from its plane
            auto A0 = [&](int sb, int kb) -> const f32x16 & { return act[sb][kb]; };
            auto A2 = [&](int sb, int kb) -> const f32x16 & { return act[sb][(2 + kb) % NFB]; };
            f32x16 gp[2][PB];
            auto reload = [&](f32x16 (*dst)[NFB]) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb)
                        dst[sb][fb] = load_block(gplane(D.g_dy(5)) + (row0 + 32 * sb) * FP, fb, i, h);
            };
            const char *w = lds + pipe.acquire();
            if constexpr (NFB == 2) {
                f32x16 y5[2][NFB];
                reload(y5);
                auto Y5 = [&](int sb, int kb) -> const f32x16 & { return y5[sb][kb]; };
                if constexpr (PB <= 2) {   // pass of 2 blocks: one chunk = dY0's two k-blocks, then dY5's
                    auto B = [&](int sb, int kb) -> const f32x16 & { return kb < 2 ? act[sb][kb < 2 ? kb : 0] : y5[sb][kb >= 2 ? kb - 2 : 0]; };
                    mma_slots2<PB, 4, 2, 0, 16, true>(gp[0], gp[1], B, w, offq, &pipe);
                } else {                   // pass of 4 blocks: two k-blocks per chunk
                    mma_slots2<PB, 2, 4, 0, 16, true>(gp[0], gp[1], A0, w, offq, &pipe);
                    mma_slots2<PB, 2, 4>(gp[0], gp[1], Y5, w + CHUNK_BYTES, offq);
                }
                pipe.issue_done();
            } else if constexpr (PB <= 2) {   // pass of 2 blocks: four k-blocks per chunk
                mma_slots2<PB, 4, 2, 0, 16, true>(gp[0], gp[1], A0, w, offq, &pipe);
                reload(act);
                mma_slots2<PB, 4, 2>(gp[0], gp[1], A0, w + CHUNK_BYTES, offq);
                pipe.issue_done();
            } else {         // pass of 4 blocks: two k-blocks per chunk, two pairs
                mma_slots2<PB, 2, 4, 0, 16, true>(gp[0], gp[1], A0, w, offq, &pipe);
                mma_slots2<PB, 2, 4>(gp[0], gp[1], A2, w + CHUNK_BYTES, offq);
                pipe.issue_done();
                reload(act);
                w = lds + pipe.acquire();
                mma_slots2<PB, 2, 4, 0, 16>(gp[0], gp[1], A0, w, offq, &pipe);
                mma_slots2<PB, 2, 4>(gp[0], gp[1], A2, w + CHUNK_BYTES, offq);
                pipe.issue_done();
            }
            save_plane<PB, true>(gplane(D.g_gp()), 32 * PB, m[0], h, gp[0]);
            save_plane<PB, true>(gplane(D.g_gp()), 32 * PB, m[1], h, gp[1]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The reverse chain of the 256-feature networks of this family (feat_dim 225..256: NSB = 1, NFB = 8) when the caller does
// not ask for input gradients -- the training step of the yaml-reachable encoder settings (coord_encode_level 11..16,
// dir_encode_level 5..10; round 5).  Same stream (dx_pass without the input stages: fc_9^T, fc_8^T .. fc_1^T, every pass
// eight output blocks, one k-block per chunk), same planes, same dW path as the general chain; but dY(l) stays in
// registers between layers (128 accumulator + 128 activation registers, the ReLU decisions as four mask words) and its
// 32 plane stores leave between the MFMA groups of the layer that multiplies it (PlaneStore: 16 in the first chunk of
// each of the layer's first two pairs), like the fused family's dX chain.  The pass program does not depend on pos_dim /
// view_dir_dim without the input stages: one instantiation.
__global__ __launch_bounds__(256, 1) void reg_dx_kernel(const WideArgs a) {
    constexpr int NFB = 8, HB = 4, FP = 256, HP = 128;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float *cb = reinterpret_cast<float *>(lds + RING_SLOTS * CHUNK_BYTES);
    for (int e = tid; e < a.D.c_floats() / 4; e += 256)
        reinterpret_cast<f32x4 *>(cb)[e] = reinterpret_cast<const f32x4 *>(a.consts)[e];
    int offq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) offq[q] = chunk_slot_offset(i, 2 * q + h);
    Pipe pipe;
    pipe.src_wave = a.stream + wave * 8192;
    pipe.lane_off = (unsigned)lane * 16u;
    pipe.lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 8192u;
    pipe.issued = 0; pipe.issue_pos = 0; pipe.consumed = 0;
    pipe.n_pairs = a.n_pairs; pipe.skip_mask = 0;
    __syncthreads();
    pipe.issue();
    const Dims &D = a.D;
    const int64_t MP = a.MP;
    auto gplane = [&](int off) { return a.grad + (int64_t)off * MP; };

    for (int64_t tile = blockIdx.x; tile < MP / 128; tile += gridDim.x) {
        const int64_t m = tile * 128 + wave * 32 + i;
        // ---- heads (nerf.py:115, :119) and dY9 = (W_out^T d y10) . [h9 > 0]
        float dsig;
        f32x16 d9[HB];
        {
            const bool valid = m < a.M;
            const int64_t mc = valid ? m : a.M - 1;
            float gy[3], yv[3], gv[3];      // (all loads first, from the clamped row)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { yv[ch] = a.rgb_in[3 * mc + ch]; gv[ch] = a.g_rgb[3 * mc + ch]; }
            const float sg = a.sigma_in[mc], gsg = a.g_sigma[mc];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float t = gv[ch] * yv[ch] * (1.0f - yv[ch]);
                gy[ch] = valid ? t : 0.0f;
            }
            dsig = (valid && sg > 0.0f) ? gsg : 0.0f;
            if (h == 0) {
                const f32x4 g4 = {gy[0], gy[1], gy[2], 0.0f};
                *reinterpret_cast<f32x4 *>(gplane(D.g_gy()) + 4 * m) = g4;
                gplane(D.g_dsig())[m] = dsig;
            }
            unsigned b9[HB / 2];      // [h9 > 0]
#pragma unroll
            for (int k = 0; k < HB / 2; ++k) b9[k] = mask_words_of(a.rec, D.r_mask(8), MP, m, h, NFB / 2, 0)[k];
#pragma unroll
            for (int fb = 0; fb < HB; ++fb) {
                f32x16 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k0 = 32 * fb + 8 * q + 4 * h;
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(cb + D.c_wout() + k0);
                    const f32x4 w1 = *reinterpret_cast<const f32x4 *>(cb + D.c_wout() + HP + k0);
                    const f32x4 w2 = *reinterpret_cast<const f32x4 *>(cb + D.c_wout() + 2 * HP + k0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[4 * q + j] = fmaf(w2[j], gy[2], fmaf(w1[j], gy[1], w0[j] * gy[0]));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) d9[fb][r] = keep_bit(b9[fb >> 1], 16 * (fb & 1) + r, v[r]);
            }
            save_plane<HB, true>(gplane(D.g_dy9()), HP, m, h, d9);
        }
        f32x16 acc[NFB], act[NFB];
        unsigned mk[NFB / 2];        // the ReLU decisions of the next seam, from the record's bit planes
        // ---- d y8[1:] = W9[:, :F]^T dY9: four k-blocks, one per chunk
        {
            const char *w = nullptr;
            auto D9 = [&](int, int kb) -> const f32x16 & { return d9[kb]; };
            static_for<HB>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if (c % 2 == 0) w = lds + pipe.acquire();
                auto b = [&](int sb, int) -> const f32x16 & { return D9(sb, c); };
                if constexpr (c % 2 == 0) mma_slots2<NFB, 1, NFB, 0, 16, c == 0, 1>(acc, acc, b, w, offq, &pipe);
                else mma_slots2<NFB, 1, NFB, 0, 0, false, 1>(acc, acc, b, w + CHUNK_BYTES, offq);
                if (c % 2 == 1) pipe.issue_done();
            });
        }
        auto load_masks = [&](int l) {
#pragma unroll
            for (int k = 0; k < NFB / 2; ++k) mk[k] = mask_words_of(a.rec, D.r_mask(l), MP, m, h, NFB / 2, 0)[k];
        };
        auto apply_masks = [&]() {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) act[fb][r] = keep_bit(mk[fb >> 1], 16 * (fb & 1) + r, acc[fb][r]);
        };
        // one layer of the chain: dY(l) = `act` is stored while it multiplies W_l^T; FRESH: the accumulators start from zero
        auto layer = [&](int l, auto fresh_tag) {
            constexpr bool FRESH = decltype(fresh_tag)::value;
            PlaneStore st;            // dY(l) leaves between the MFMA groups of the first chunks of the first two pairs
            st.open(gplane(D.g_dy(l)), FP, m, h, act);
            const char *w = nullptr;
            static_for<NFB>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if (c % 2 == 0) w = lds + pipe.acquire();
                auto b = [&](int, int) -> const f32x16 & { return act[c]; };
                if constexpr (c == 0) mma_slots2<NFB, 1, NFB, 0, 16, FRESH, 1, 16, 32, 32>(acc, acc, b, w, offq, &pipe, &st, 0);
                else if constexpr (c == 2) mma_slots2<NFB, 1, NFB, 0, 16, false, 1, 16, 32, 32>(acc, acc, b, w, offq, &pipe, &st, 16);
                else if constexpr (c % 2 == 0) mma_slots2<NFB, 1, NFB, 0, 16, false, 1>(acc, acc, b, w, offq, &pipe);
                else mma_slots2<NFB, 1, NFB, 0, 0, false, 1>(acc, acc, b, w + CHUNK_BYTES, offq);
                if (c % 2 == 1) pipe.issue_done();
            });
        };
        // ---- l = 8: dY7' = W8[1:]^T d y8 + W8[0, :] dsigma' (the density row); fc_8 has no ReLU: d y8 is the accumulators as they are
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) act[fb] = acc[fb];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(cb + D.c_w8row() + 32 * fb + 8 * q + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[fb][4 * q + j] = wv[j] * dsig;
            }
        load_masks(7);
        layer(8, std::false_type());
        // ---- l = 7 .. 1: dY(l-1)' = W_l^T dY(l), dY(l) = dY(l)' . [h(l) > 0]
        for (int l = 7; l >= 1; --l) {
            apply_masks();
            load_masks(l - 1);        // masks of the next seam, fetched one stage ahead
            layer(l, std::true_type());
        }
        // ---- dY0
        apply_masks();
        save_plane<NFB, true>(gplane(D.g_dy(0)), FP, m, h, act);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int launch_reg_dx(const WideArgs &a, hipStream_t s) {
    static nerf::DeviceMask configured{0};
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(reg_dx_kernel), REG_LDS, configured,
                                          "nerf_mlp_layered: LDS attribute (register-resident reverse chain)"))
        return rc;
    const int64_t ntiles = a.MP / 128;
    const int cus = nerf::device_cus();
    hipLaunchKernelGGL(reg_dx_kernel, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(256), REG_LDS, s, a);
    return nerf::check_launch("nerf_mlp_layered_backward: reverse chain (register-resident, 256 features)");
}

template <int NSB, int NFB, int PB, int DB = 1>
int launch_reg_fwd(bool record, const WideArgs &a, hipStream_t s) {
    auto kern = record ? reg_forward_kernel<NSB, NFB, PB, true, DB> : reg_forward_kernel<NSB, NFB, PB, false, DB>;
    static nerf::DeviceMask configured[2] = {{0}, {0}};
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), REG_LDS, configured[record],
                                          "nerf_mlp_layered: LDS attribute (register-resident forward)"))
        return rc;
    const int64_t ntiles = a.MP / (128 * NSB);
    const int cus = nerf::device_cus();
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(256), REG_LDS, s, a);
    return nerf::check_launch("nerf_mlp_layered_forward (register-resident)");
}
int launch_reg_forward(bool record, const WideArgs &a, hipStream_t s) {
    const int pb = a.D.Pp / 32;
    if (a.D.Fp == 64)
        return pb == 1 ? launch_reg_fwd<2, 2, 1>(record, a, s) : pb == 2 ? launch_reg_fwd<2, 2, 2>(record, a, s)
             : pb == 3 ? launch_reg_fwd<2, 2, 3>(record, a, s) : launch_reg_fwd<2, 2, 4>(record, a, s);
    if (a.D.Fp == 128)
        return pb == 1 ? launch_reg_fwd<2, 4, 1>(record, a, s) : pb == 2 ? launch_reg_fwd<2, 4, 2>(record, a, s) : launch_reg_fwd<2, 4, 3>(record, a, s);
    if (a.D.Dp == 64)
        return pb == 1 ? launch_reg_fwd<1, 8, 1, 2>(record, a, s) : pb == 2 ? launch_reg_fwd<1, 8, 2, 2>(record, a, s)
             : pb == 3 ? launch_reg_fwd<1, 8, 3, 2>(record, a, s) : launch_reg_fwd<1, 8, 4, 2>(record, a, s);
    return pb == 1 ? launch_reg_fwd<1, 8, 1>(record, a, s) : pb == 2 ? launch_reg_fwd<1, 8, 2>(record, a, s)
         : pb == 3 ? launch_reg_fwd<1, 8, 3>(record, a, s) : launch_reg_fwd<1, 8, 4>(record, a, s);
}

template <int NFB, int PB>
int launch_narrow_dx_pb(bool ig, const WideArgs &a, hipStream_t s) {
    auto kern = ig ? narrow_dx_kernel<NFB, PB, true> : narrow_dx_kernel<NFB, PB, false>;
    static nerf::DeviceMask configured[2] = {{0}, {0}};
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), REG_LDS, configured[ig],
                                          "nerf_mlp_layered: LDS attribute (narrow reverse chain)"))
        return rc;
    const int64_t ntiles = a.MP / 256;
    const int cus = nerf::device_cus();
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(256), REG_LDS, s, a);
    return nerf::check_launch("nerf_mlp_layered_backward: reverse chain (narrow)");
}
int launch_narrow_dx(bool ig, const WideArgs &a, hipStream_t s) {
    const int pb = a.D.Pp / 32;
    if (a.D.Fp == 64)
        return pb == 1 ? launch_narrow_dx_pb<2, 1>(ig, a, s) : pb == 2 ? launch_narrow_dx_pb<2, 2>(ig, a, s)
             : pb == 3 ? launch_narrow_dx_pb<2, 3>(ig, a, s) : launch_narrow_dx_pb<2, 4>(ig, a, s);
    return pb == 1 ? launch_narrow_dx_pb<4, 1>(ig, a, s) : pb == 2 ? launch_narrow_dx_pb<4, 2>(ig, a, s) : launch_narrow_dx_pb<4, 3>(ig, a, s);
}

// thin reductions of the reverse pass (vector ALU, HBM-bound: one more read of the h7 and h9 planes):
//   fc_8.weight[0, k] = sum_m dsig[m] h7[m][k],  fc_out.weight[c][k] = sum_m gy[m][c] h9[m][k],
//   fc_8.bias[0] = sum dsig,  fc_out.bias[c] = sum gy[.][c]
// One wavefront per (32-feature block, slice of the sample axis): a 32-sample tile of the block is 4 KiB = four
// coalesced 16-byte loads per lane; lane (i, h) keeps the partial sums of sample i of every tile in double, the 32
// lanes of a half are added up once at the end.  Slices are summed in a fixed order by the second kernel: no atomics.
constexpr int THIN_SLICES = 1024;   // (256 single-wavefront slices per block left the kernel latency-bound at 2.4 TB/s)
// scalars_only (feat_dim 256: the rows are side jobs of the dW list kernel): two blocks per slice, the sums of dsig and of
// gy only -- the four bias gradients -- without touching the h7 / h9 planes
__global__ __launch_bounds__(64) void layered_thin_kernel(const Dims D, const float *__restrict__ rec,
                                                          const float *__restrict__ grad, int64_t MP, int slices,
                                                          double *__restrict__ partial, int scalars_only) {
    const int cols = D.Fp + 3 * D.Hp + 4;
    const int blk = blockIdx.x, lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const bool is_h7 = scalars_only ? blk == 0 : blk < D.Fp / 32;
    const int fb = scalars_only ? 0 : is_h7 ? blk : blk - D.Fp / 32;
    const int width = is_h7 ? D.Fp : D.Hp;
    const float *plane = rec + (int64_t)(is_h7 ? D.r_h(7) : D.r_h9()) * MP;
    const float *gy = grad + (int64_t)D.g_gy() * MP, *ds = grad + (int64_t)D.g_dsig() * MP;
    const int64_t tiles = MP / 32, per = (tiles + slices - 1) / slices;
    const int64_t t0 = blockIdx.y * per, t1 = t0 + per < tiles ? t0 + per : tiles;
    double acc[3][16], ssum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0;
    constexpr int TPT = 8;
    // TPT tiles per trip: their 4 TPT loads are in flight together (one tile at a time, each wavefront sat through a
    // loaded HBM round trip per 4 KiB: 2.4 TB/s with 12 wavefronts per CU)
    for (int64_t t = t0; t < t1; t += TPT) {
        f32x16 x[TPT];
        float g1[TPT];
        f32x4 g4[TPT];
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const int64_t tj = t + j < t1 ? t + j : t1 - 1;       // past the end: a re-read with zero weights
            const int64_t m = tj * 32 + i;
            if (!scalars_only) x[j] = load_block(plane + tj * 32 * width, fb, i, h);
            if (is_h7) g1[j] = t + j < t1 ? ds[m] : 0.0f;
            else {
                g4[j] = *reinterpret_cast<const f32x4 *>(gy + 4 * m);
                if (t + j >= t1) g4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            if (is_h7) {
                const double g = (double)g1[j];
                if (h == 0) ssum[0] += g;
                if (scalars_only) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += g * (double)x[j][r];
            } else {
                if (h == 0) { ssum[1] += (double)g4[j].x; ssum[2] += (double)g4[j].y; ssum[3] += (double)g4[j].z; }
                if (scalars_only) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[0][r] += (double)g4[j].x * (double)x[j][r];
                    acc[1][r] += (double)g4[j].y * (double)x[j][r];
                    acc[2][r] += (double)g4[j].z * (double)x[j][r];
                }
            }
        }
    }
    auto over_samples = [&](double v) {   // sum over the 32 lanes of this half
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
        return v;
    };
    double *out = partial + (int64_t)blockIdx.y * cols;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if ((is_h7 && c > 0) || scalars_only) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double v = over_samples(acc[c][r]);
            const int k = 32 * fb + (r & 3) + 8 * (r >> 2) + 4 * h;      // the feature register r of lane half h holds
            if (i == 0) out[is_h7 ? k : D.Fp + c * D.Hp + k] = v;
        }
    }
    if (fb == 0) {   // the scalar sums ride on the first block of each plane (lane half 0 holds them)
        if (is_h7) { const double v = over_samples(ssum[0]); if (lane == 0) out[D.Fp + 3 * D.Hp] = v; }
        else {
#pragma unroll
            for (int c = 1; c < 4; ++c) { const double v = over_samples(ssum[c]); if (lane == 0) out[D.Fp + 3 * D.Hp + c] = v; }
        }
    }
}
__global__ __launch_bounds__(64) void layered_thin_reduce_kernel(const Dims D, const double *__restrict__ partial,
                                                                 int slices, float *__restrict__ g_params, int first_col) {
    const int cols = D.Fp + 3 * D.Hp + 4;
    const int col = first_col + blockIdx.x;
    __shared__ double part[64];
    double acc = 0.0;
    for (int z = threadIdx.x; z < slices; z += 64) acc += partial[(int64_t)z * cols + col];   // fixed order per thread
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int t = 1; t < 64; ++t) acc += part[t];                                                // and across threads
    const float s = (float)acc;
    if (col < D.Fp) { if (col < D.F) g_params[D.w[8] + col] = s; }
    else if (col < D.Fp + 3 * D.Hp) {
        const int ch = (col - D.Fp) / D.Hp, k = (col - D.Fp) % D.Hp;
        if (k < D.H) g_params[D.w[10] + ch * D.H + k] = s;
    } else {
        const int ch = col - D.Fp - 3 * D.Hp;
        g_params[ch == 0 ? D.b[8] : D.b[10] + ch - 1] = s;
    }
}

inline int64_t align256b(int64_t b) { return (b + 255) & ~(int64_t)255; }

// buffer layouts (bytes)
struct Sizes {
    int64_t consts, fwd_stream, dx_stream, planes;   // planes: record (forward) or gradient planes (backward)
};
Sizes sizes(const Dims &D, int64_t rows, bool backward, int inputs) {
    Sizes s;
    s.consts = align256b(4 * (int64_t)D.c_floats());
    s.fwd_stream = (int64_t)stream_pairs(D, 0, 0) * PAIR_BYTES;
    s.dx_stream = backward ? (int64_t)stream_pairs(D, 1, inputs) * PAIR_BYTES : 0;
    s.planes = align256b(4 * lrows(rows) * (int64_t)(backward ? D.gradw() : D.recw()));
    return s;
}

#ifdef X_LAYERED_TIMELINE
extern "C" __attribute__((visibility("default"))) int nerf_debug_layered_timeline(unsigned long long *out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lt), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lt), z, sizeof(z)); }
    return 8;
}
#endif

int launch_program(bool dx, const WideArgs &a, hipStream_t s) {
    auto kern = dx ? layered_kernel<true> : layered_kernel<false>;
    static nerf::DeviceMask configured[2] = {{0}, {0}};
    constexpr int GEN_LDS = RING_SLOTS * CHUNK_BYTES + PASS_TABLE_MAX * PASS_ROW * 4;   // ring + the tabulated pass program
    if (int rc = nerf::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), GEN_LDS, configured[dx],
                                          "nerf_mlp_layered: LDS attribute"))
        return rc;
    const int64_t ntiles = a.MP / TILE_SAMPLES;
    const int cus = nerf::device_cus();
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(256), GEN_LDS, s, a);
    return nerf::check_launch(dx ? "nerf_mlp_layered_backward: reverse chain" : "nerf_mlp_layered_forward");
}

unsigned grid_for(int64_t total) { return (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }

}  // namespace

// record buffer = [constant block][forward stream][planes of `rows` rows]
NERF_API int64_t nerf_mlp_layered_record_bytes(const nerf_net_t *net, int64_t rows) {
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0) return -1;
    if (rows <= 0) return 0;
    const Dims D = make_dims(d);
    const Sizes z = sizes(D, rows, false, 0);
    return z.consts + z.fwd_stream + z.planes;
}

// Where a plane of the record sits (for tools and tests; the record is otherwise opaque): which = 0 encoded position,
// 1 encoded direction, 2..9 h0..h7 (post-ReLU), 10 fc_8 rows 1..F (no ReLU), 11 h9.  Returns the byte offset inside a
// record of `rows` rows and the plane's padded width (floats per sample, tile-fragment layout: nerf_mlp_plane_offset).
NERF_API int64_t nerf_mlp_layered_plane(const nerf_net_t *net, int64_t rows, int which, int *width) {
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0 || rows <= 0 || which < 0 || which > 11) return -1;
    const Dims D = make_dims(d);
    const Sizes z = sizes(D, rows, false, 0);
    const int off = which == 0 ? D.r_pe() : which == 1 ? D.r_de() : which == 11 ? D.r_h9() : D.r_h(which - 2);
    if (width) *width = which == 0 ? D.Pp : which == 1 ? D.Dp : which == 11 ? D.Hp : D.Fp;
    return z.consts + z.fwd_stream + 4 * (int64_t)off * lrows(rows);
}

// How many <= 256 x 256 windows the dW pass cuts the eleven layers into: the same enumeration as the `add` calls of
// nerf_mlp_layered_backward (windows of a (n_w x x_w) product: ceil(n_w / 256) * ceil(x_w / 256)), so that the
// workspace is sized for the list that will really be built -- also for very wide pos_dim / view_dir_dim.
static int dw_item_budget(const Dims &D) {
    auto win = [](int n_w, int x_w) { return ((n_w + 255) / 256) * ((x_w + 255) / 256); };
    int n = win(D.Fp, D.Pp);                                   // fc_in
    for (int l = 1; l <= 8; ++l) n += win(D.Fp, D.Fp) + (l == 5 ? win(D.Fp, D.Pp) : 0);
    return n + win(D.Hp, D.Fp) + win(D.Hp, D.Dp);              // fc_9: features, then the direction
}

// workspace = [reverse stream (with the input-gradient passes)][gradient planes][thin partials][dW scratch]
NERF_API int64_t nerf_mlp_layered_workspace_bytes(const nerf_net_t *net, int64_t M) {
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0) return -1;
    if (M <= 0) return 0;
    const Dims D = make_dims(d);
    const Sizes z = sizes(D, M, true, 1);
    const int64_t thin = align256b(8 * (int64_t)THIN_SLICES * (D.Fp + 3 * D.Hp + 4));
    return z.dx_stream + z.planes + thin + nerf::dw_items_scratch_bytes(dw_item_budget(D)) + 65536;
}

// The dW / db work list of one backward call: windows of <= 256 x 256 over (dY plane of the layer, its input plane(s)).
// Pure pointer arithmetic on the three bases -- nerf_mlp_layered_plan_check runs it with stand-in bases, no GPU.
// feat_dim 256: the density row of fc_8 and fc_out.weight are summed beside the fc_8 / fc_9 GEMMs (the fused family's side
// jobs, mlp_backward.hip:dw_body) from tiles that are in LDS anyway, instead of a separate pass over the h7 and h9 planes
// (1.2 GB per fine pass, 0.49 ms of a 27.8 ms step: profiles/r05_encoders_train_trace.txt)
static bool dw_side_jobs(const Dims &D) { return D.F == 256; }
static const float *rp_plane(const float *buffer, int off, int64_t MP) { return buffer + (int64_t)off * MP; }

static void enumerate_dw_items(const Dims &D, int64_t MP, float *g_params, const float *rec, const float *grad,
                               std::vector<nerf::DwItem> &items) {
    const bool side_jobs = dw_side_jobs(D);
    auto add = [&](int layer, const float *dy_plane, int n_w, int rows_total, int row0, const float *x_plane, int x_w,
                   int cols_total, int col0, bool bias, int side = 0) {
        for (int fa = 0; fa * 32 < n_w; fa += 8)
            for (int fx = 0; fx * 32 < x_w; fx += 8) {
                nerf::DwItem it;
                it.a_plane = dy_plane; it.a_width = n_w; it.a_fb0 = fa; it.a_blocks = n_w / 32 - fa < 8 ? n_w / 32 - fa : 8;
                it.x_plane = x_plane; it.x_width = x_w; it.x_fb0 = fx; it.x_blocks = x_w / 32 - fx < 8 ? x_w / 32 - fx : 8;
                it.ld = D.in[layer];
                // (an item with a side job keeps row0 OUT of its destinations: the reducer adds it, and the density row
                // lands at the tensor's start)
                it.side = side; it.row0 = side ? row0 : 0;
                const int r0 = side ? 0 : row0;
                it.w_dst = g_params + D.w[layer] + (int64_t)(r0 + fa * 32) * D.in[layer] + col0 + fx * 32;
                it.rows_valid = rows_total - fa * 32 < 256 ? rows_total - fa * 32 : 256;
                it.cols_valid = cols_total - fx * 32 < 256 ? cols_total - fx * 32 : 256;
                it.b_dst = (bias && fx == 0) ? g_params + D.b[layer] + r0 + fa * 32 : nullptr;
                if (it.rows_valid > 0 && it.cols_valid > 0) items.push_back(it);
            }
    };
    auto gp = [&](int off) { return grad + (int64_t)off * MP; };
    auto rp = [&](int off) { return rec + (int64_t)off * MP; };
    add(0, gp(D.g_dy(0)), D.Fp, D.F, 0, rp(D.r_pe()), D.Pp, D.E_p, 0, true);
    for (int l = 1; l <= 8; ++l) {
        if (l == 5) add(5, gp(D.g_dy(5)), D.Fp, D.F, 0, rp(D.r_pe()), D.Pp, D.E_p, 0, false);
        add(l, gp(D.g_dy(l)), D.Fp, D.F, l == 8 ? 1 : 0, rp(D.r_h(l - 1)), D.Fp, D.F, l == 5 ? D.E_p : 0, true,
            (l == 8 && side_jobs) ? 2 : 0);
    }
    add(9, gp(D.g_dy9()), D.Hp, D.H, 0, rp(D.r_h(8)), D.Fp, D.F, 0, true, side_jobs ? 4 : 0);
    add(9, gp(D.g_dy9()), D.Hp, D.H, 0, rp(D.r_de()), D.Dp, D.E_d, D.F, false);
}

// Host-only dry run of the backward call's bookkeeping for NeRF(net) on M samples and a device of `cus` compute units
// (<= 0: 256, the MI355X): the workspace layout, the dW work list against dw_item_budget, every item's destination
// rectangle against the parameter blob, every item's READ extent against the record / workspace it points into, and
// the list's partial-tile buffer against what the workspace reserves for it.  Nothing is launched and no pointer is
// dereferenced: tests/test_layered_plan.py sweeps it on the CPU (under the sanitizers too).
NERF_API int nerf_mlp_layered_plan_check(const nerf_net_t *net, int64_t M, int cus) {
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0) return NERF_ERR_ARG;
    NERF_REQUIRE(M >= 0 && M < (int64_t)1 << 31, "nerf_mlp_layered_plan_check: M out of range");
    if (M == 0) return NERF_OK;
    if (cus <= 0) cus = 256;
    const Dims D = make_dims(d);
    const int64_t MP = lrows(M);
    const Sizes zr = sizes(D, M, false, 0), zw = sizes(D, M, true, 1);
    const int64_t record_bytes = nerf_mlp_layered_record_bytes(net, M), workspace_bytes = nerf_mlp_layered_workspace_bytes(net, M);
    const int64_t thin_bytes = align256b(8 * (int64_t)THIN_SLICES * (D.Fp + 3 * D.Hp + 4));
    const int64_t dw_scratch_off = zw.dx_stream + zw.planes + thin_bytes;
    const int64_t dw_scratch_bytes = nerf::dw_items_scratch_bytes(dw_item_budget(D)) + 65536;
    char msg[256];
#define PLAN_REQUIRE(cond, ...)                                     \
    do {                                                            \
        if (!(cond)) {                                              \
            snprintf(msg, sizeof msg, __VA_ARGS__);                 \
            return nerf::fail(NERF_ERR_ARG, msg);                   \
        }                                                           \
    } while (0)
    PLAN_REQUIRE(record_bytes == zr.consts + zr.fwd_stream + zr.planes && zr.planes >= 4 * MP * (int64_t)D.recw(),
                 "plan: record of %lld bytes does not hold %lld rows of %d floats", (long long)record_bytes, (long long)MP, D.recw());
    PLAN_REQUIRE(zw.planes >= 4 * MP * (int64_t)D.gradw(), "plan: gradient planes");
    PLAN_REQUIRE(dw_scratch_off + dw_scratch_bytes <= workspace_bytes, "plan: workspace of %lld bytes ends before the dW scratch (%lld + %lld)",
                 (long long)workspace_bytes, (long long)dw_scratch_off, (long long)dw_scratch_bytes);
    // stand-in bases, far apart: only differences are ever formed
    char *const RECORD = reinterpret_cast<char *>((uintptr_t)1 << 44), *const WORK = reinterpret_cast<char *>((uintptr_t)2 << 44);
    float *const PARAMS = reinterpret_cast<float *>((uintptr_t)3 << 44);
    const float *rec = reinterpret_cast<const float *>(RECORD + zr.consts + zr.fwd_stream);
    const float *grad = reinterpret_cast<const float *>(WORK + zw.dx_stream);
    std::vector<nerf::DwItem> items;
    enumerate_dw_items(D, MP, PARAMS, rec, grad, items);
    PLAN_REQUIRE((int)items.size() <= dw_item_budget(D), "plan: %d dW windows, workspace sized for %d", (int)items.size(), dw_item_budget(D));
    int64_t covered = 0;
    for (size_t k = 0; k < items.size(); ++k) {
        const nerf::DwItem &it = items[k];
        // destination rectangle inside ONE layer's weight tensor
        const int64_t w0 = it.w_dst - PARAMS;
        int layer = -1;
        for (int l = 0; l < 10; ++l)
            if (w0 >= D.w[l] && w0 < D.w[l] + (int64_t)D.in[l] * D.out[l]) layer = l;
        PLAN_REQUIRE(layer >= 0 && it.ld == D.in[layer], "plan: item %d writes outside every weight tensor", (int)k);
        const int64_t rel = w0 - D.w[layer], row = rel / it.ld + it.row0, col = rel % it.ld;
        PLAN_REQUIRE(it.rows_valid >= 1 && it.rows_valid <= 256 && it.cols_valid >= 1 && it.cols_valid <= 256 &&
                     row + it.rows_valid <= D.out[layer] && col + it.cols_valid <= D.in[layer],
                     "plan: item %d (layer %d) writes rows %lld+%d of %d, columns %lld+%d of %d", (int)k, layer, (long long)row,
                     it.rows_valid, D.out[layer], (long long)col, it.cols_valid, D.in[layer]);
        covered += (int64_t)it.rows_valid * it.cols_valid;
        if (it.b_dst) {
            const int64_t b0 = it.b_dst - PARAMS;
            PLAN_REQUIRE(b0 + it.row0 >= D.b[layer] && b0 + it.row0 + it.rows_valid <= D.b[layer] + D.out[layer], "plan: item %d bias rows", (int)k);
        }
        if (it.side) {      // side jobs: the 256 x 256 fc_8 item / the 128 x 256 fc_9 item of a 256-feature network, nothing else
            PLAN_REQUIRE(D.F == 256 && ((it.side == 2 && layer == 8 && it.a_blocks == 8 && it.x_blocks == 8 && it.row0 == 1 && rel == 0) ||
                                        (it.side == 4 && layer == 9 && it.a_blocks == 4 && it.x_blocks == 8 && it.row0 == 0 && rel == 0)),
                         "plan: item %d carries side job %d it is not shaped for", (int)k, it.side);
        }
        PLAN_REQUIRE(it.a_blocks >= 1 && it.a_blocks <= 8 && it.x_blocks >= 1 && it.x_blocks <= 8 &&
                     it.a_blocks * 32 >= it.rows_valid && it.x_blocks * 32 >= it.cols_valid, "plan: item %d window blocks", (int)k);
        // what the dW kernel reads: inside the buffer the plane lives in
        int64_t a_floats = 0, x_floats = 0;
        nerf::dw_item_read_extent(it, M, &a_floats, &x_floats);
        const int64_t a_end = (reinterpret_cast<const char *>(it.a_plane) - WORK) + 4 * a_floats;
        const int64_t x_end = (reinterpret_cast<const char *>(it.x_plane) - RECORD) + 4 * x_floats;
        PLAN_REQUIRE(a_end <= workspace_bytes, "plan: item %d reads its dY window up to byte %lld of a %lld-byte workspace", (int)k,
                     (long long)a_end, (long long)workspace_bytes);
        PLAN_REQUIRE(x_end <= record_bytes, "plan: item %d reads its X window up to byte %lld of a %lld-byte record", (int)k,
                     (long long)x_end, (long long)record_bytes);
    }
    // every element of the ten weight tensors (fc_8 without its density row, which the thin kernel owns) exactly once
    int64_t want = 0;
    for (int l = 0; l < 10; ++l) want += (int64_t)D.in[l] * (l == 8 ? D.out[l] - 1 : D.out[l]);
    PLAN_REQUIRE(covered == want, "plan: the windows cover %lld weight elements of %lld", (long long)covered, (long long)want);
    const int64_t need = nerf::dw_items_needed_bytes(items, M, cus);
    PLAN_REQUIRE(need <= dw_scratch_bytes, "plan: the dW list needs %lld bytes of partial tiles, the workspace reserves %lld", (long long)need,
                 (long long)dw_scratch_bytes);
#undef PLAN_REQUIRE
    return NERF_OK;
}

NERF_API int nerf_mlp_layered_forward(const nerf_net_t *net, const float *params, const float *pos,
                                      const float *view_dir, int64_t M, int encoded, float *sigma, float *rgb,
                                      void *record, int64_t record_rows, int keep_record, nerf_stream_t stream) {
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0) return NERF_ERR_ARG;
    NERF_REQUIRE(M >= 0, "nerf_mlp_layered_forward: negative M");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(params && pos && view_dir && sigma && rgb && record && record_rows > 0,
                 "nerf_mlp_layered_forward: null pointer");
    NERF_REQUIRE(M < (int64_t)1 << 31, "nerf_mlp_layered_forward: more than 2^31 samples per call");
    NERF_REQUIRE(!keep_record || record_rows >= M, "nerf_mlp_layered_forward: a kept record needs record_rows >= M");
    const bool recording = keep_record != 0;
    const Dims D = make_dims(d);
    if (!encoded) {   // RAW (M, 3) points / directions: the two PositionalEncoders of nerf_net_t are applied on the way into the planes
        if (d.pos_levels < 0 || d.dir_levels < 0)
            return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_mlp_layered_forward: raw inputs need the levels of both PositionalEncoders "
                                                    "in nerf_net_t (other encoders: encode first, pass encoded = 1)");
        NERF_REQUIRE(D.Pp <= ENC_MAX_W && D.Dp <= ENC_MAX_W, "nerf_mlp_layered_forward: raw inputs: encodings wider than 256");
    }
    hipStream_t s = nerf::as_stream(stream);
    int64_t chunk = record_rows < M ? record_rows : M;
    const Sizes z = sizes(D, chunk, false, 0);
    // an inference call on a network that fits the register file touches only the two input planes of its scratch:
    // the same bytes hold recw / (Pp + Dp) times the rows (fewer, longer launches: 12+ tiles per CU instead of one)
    if (reg_ok(D) && !recording) {
        const int64_t fit = lrows(record_rows) * D.recw() / (D.Pp + D.Dp) / 256 * 256;
        chunk = fit < M ? fit : M;
    }
    char *base = static_cast<char *>(record);
    float *consts = reinterpret_cast<float *>(base);
    float *fstream = reinterpret_cast<float *>(base + z.consts);
    float *planes = reinterpret_cast<float *>(base + z.consts + z.fwd_stream);
    // pack (once per call: the parameters may have changed; 2 x the parameter bytes of HBM traffic)
    hipLaunchKernelGGL(layered_pack_consts, dim3(grid_for(D.c_floats())), dim3(256), 0, s, D, params, consts);
    PackArgs pa; pa.D = D; pa.P = params; pa.dx = 0; pa.inputs = 0; pa.n_pairs = stream_pairs(D, 0, 0);
    hipLaunchKernelGGL(layered_pack_stream, dim3((unsigned)(pa.n_pairs < 256 ? 16 * pa.n_pairs : 4096)), dim3(256), 0, s, pa, fstream);
    if (int rc = nerf::check_launch("nerf_mlp_layered_forward: pack")) return rc;
    for (int64_t r0 = 0; r0 < M; r0 += chunk) {
        const int64_t rows = M - r0 < chunk ? M - r0 : chunk;
        const int64_t MP = lrows(rows);
        if (encoded) {
            hipLaunchKernelGGL(rows_to_plane_kernel, dim3(grid_for(MP * 8)), dim3(256), 0, s, pos + r0 * D.E_p, rows, MP,
                               D.E_p, D.Pp, planes + (int64_t)D.r_pe() * MP);
            hipLaunchKernelGGL(rows_to_plane_kernel, dim3(grid_for(MP * 8)), dim3(256), 0, s, view_dir + r0 * D.E_d, rows, MP,
                               D.E_d, D.Dp, planes + (int64_t)D.r_de() * MP);
        } else {
            hipLaunchKernelGGL(encode_to_plane_kernel, dim3(grid_for(MP * 8)), dim3(256), 0, s, pos + r0 * 3, rows, MP,
                               d.pos_levels, d.pos_include_input ? 1 : 0, D.E_p, D.Pp, planes + (int64_t)D.r_pe() * MP);
            hipLaunchKernelGGL(encode_to_plane_kernel, dim3(grid_for(MP * 8)), dim3(256), 0, s, view_dir + r0 * 3, rows, MP,
                               d.dir_levels, d.dir_include_input ? 1 : 0, D.E_d, D.Dp, planes + (int64_t)D.r_de() * MP);
        }
        if (int rc = nerf::check_launch("nerf_mlp_layered_forward: input planes")) return rc;
        WideArgs a = {};
        a.D = D; a.stream = reinterpret_cast<const char *>(fstream); a.consts = consts;
        a.rec = planes; a.grad = nullptr; a.M = rows; a.MP = MP;
        a.n_passes = fwd_num_passes(D); a.n_pairs = pa.n_pairs; a.inputs = 0; a.record = recording;
        a.sigma = sigma + r0; a.rgb = rgb + 3 * r0;
        // networks that fit the register file: planes written only when the batch is recorded for a backward
        if (int rc = reg_ok(D) ? launch_reg_forward(recording, a, s) : launch_program(false, a, s)) return rc;
    }
    return NERF_OK;
}

NERF_API int nerf_mlp_layered_backward(const nerf_net_t *net, const float *params, const float *pos,
                                       const float *view_dir, int64_t M, const float *sigma, const float *rgb,
                                       const void *record, const float *g_sigma, const float *g_rgb,
                                       float *g_params, float *g_pos, float *g_view_dir, void *workspace,
                                       nerf_stream_t stream) {
    (void)pos; (void)view_dir;   // the record holds the input planes
    nerf_net_t d;
    if (nerf::net_describe(net, d) < 0) return NERF_ERR_ARG;
    NERF_REQUIRE(M >= 0, "nerf_mlp_layered_backward: negative M");
    NERF_REQUIRE(g_params, "nerf_mlp_layered_backward: null g_params");
    const Dims D = make_dims(d);
    hipStream_t s = nerf::as_stream(stream);
    if (M == 0) {
        if (hipMemsetAsync(g_params, 0, sizeof(float) * D.total, s) != hipSuccess)
            return nerf::check_launch("nerf_mlp_layered_backward: memset");
        return NERF_OK;
    }
    NERF_REQUIRE(params && sigma && rgb && record && g_sigma && g_rgb && workspace,
                 "nerf_mlp_layered_backward: null pointer");
    NERF_REQUIRE(M < (int64_t)1 << 31, "nerf_mlp_layered_backward: more than 2^31 samples per call");
    const int inputs = (g_pos || g_view_dir) ? 1 : 0;
    const int64_t MP = lrows(M);
    const Sizes zr = sizes(D, M, false, 0), zw = sizes(D, M, true, 1);
    const char *rbase = static_cast<const char *>(record);
    const float *consts = reinterpret_cast<const float *>(rbase);
    float *rec = reinterpret_cast<float *>(const_cast<char *>(rbase) + zr.consts + zr.fwd_stream);
    char *wbase = static_cast<char *>(workspace);
    float *dstream = reinterpret_cast<float *>(wbase);
    float *grad = reinterpret_cast<float *>(wbase + zw.dx_stream);
    double *thin = reinterpret_cast<double *>(wbase + zw.dx_stream + zw.planes);
    const int64_t thin_bytes = align256b(8 * (int64_t)THIN_SLICES * (D.Fp + 3 * D.Hp + 4));
    char *dw_scratch = wbase + zw.dx_stream + zw.planes + thin_bytes;

    PackArgs pa; pa.D = D; pa.P = params; pa.dx = 1; pa.inputs = inputs; pa.n_pairs = stream_pairs(D, 1, inputs);
    hipLaunchKernelGGL(layered_pack_stream, dim3((unsigned)(pa.n_pairs < 256 ? 16 * pa.n_pairs : 4096)), dim3(256), 0, s, pa, dstream);
    if (int rc = nerf::check_launch("nerf_mlp_layered_backward: pack")) return rc;
    WideArgs a = {};
    a.D = D; a.stream = reinterpret_cast<const char *>(dstream); a.consts = consts;
    a.rec = rec; a.grad = grad; a.M = M; a.MP = MP;
    a.n_passes = dx_num_passes(D, inputs); a.n_pairs = pa.n_pairs; a.inputs = inputs;
    a.sigma_in = sigma; a.rgb_in = rgb; a.g_sigma = g_sigma; a.g_rgb = g_rgb;
    // feat_dim <= 128: the narrow chains; feat_dim 225..256 without input gradients (every training step): the
    // register-resident chain; everything else the general plane-parked chain
    if (int rc = (reg_ok(D) && D.Fp <= 128) ? launch_narrow_dx(inputs != 0, a, s)
                 : (reg_ok(D) && D.Fp == 256 && !inputs) ? launch_reg_dx(a, s) : launch_program(true, a, s)) return rc;
    if (g_pos) hipLaunchKernelGGL(plane_to_rows_kernel, dim3(grid_for(M * D.E_p)), dim3(256), 0, s,
                                  grad + (int64_t)D.g_gp() * MP, M, D.E_p, D.Pp, g_pos);
    if (g_view_dir) hipLaunchKernelGGL(plane_to_rows_kernel, dim3(grid_for(M * D.E_d)), dim3(256), 0, s,
                                       grad + (int64_t)D.g_gd() * MP, M, D.E_d, D.Dp, g_view_dir);
    // thin rows
    {
        const int cols = D.Fp + 3 * D.Hp + 4;
        int slices = (int)(MP / 32 / 8);           // >= 8 tiles per slice
        if (slices > THIN_SLICES) slices = THIN_SLICES;
        if (slices < 1) slices = 1;
        if (dw_side_jobs(D)) {   // the rows ride in the dW list kernel below; only the four bias sums are left
            hipLaunchKernelGGL(layered_thin_kernel, dim3(2, slices), dim3(64), 0, s, D, rec, grad, MP, slices, thin, 1);
            hipLaunchKernelGGL(layered_thin_reduce_kernel, dim3(4), dim3(64), 0, s, D, thin, slices, g_params, cols - 4);
        } else {
            hipLaunchKernelGGL(layered_thin_kernel, dim3((D.Fp + D.Hp) / 32, slices), dim3(64), 0, s, D, rec, grad, MP, slices, thin, 0);
            hipLaunchKernelGGL(layered_thin_reduce_kernel, dim3(cols), dim3(64), 0, s, D, thin, slices, g_params, 0);
        }
        if (int rc = nerf::check_launch("nerf_mlp_layered_backward: thin rows")) return rc;
    }
    // dW / db: windows of <= 256 x 256 over (dY plane of the layer, its input plane(s))
    std::vector<nerf::DwItem> items;
    enumerate_dw_items(D, MP, g_params, rec, grad, items);
    // (the bytes really left in the workspace nerf_mlp_layered_workspace_bytes sized, not a figure derived from the list)
    NERF_REQUIRE((int)items.size() <= dw_item_budget(D), "nerf_mlp_layered_backward: more dW windows than the workspace was sized for");
    nerf::DwSide side = {rp_plane(rec, D.r_h9(), MP), rp_plane(grad, D.g_dsig(), MP), rp_plane(grad, D.g_gy(), MP), g_params + D.w[10]};
    return nerf::run_dw_items(items, M, dw_scratch, nerf::dw_items_scratch_bytes(dw_item_budget(D)) + 65536, s,
                              dw_side_jobs(D) ? &side : nullptr);
}
