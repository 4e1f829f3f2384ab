// Geometry of the fused NeRF MLP kernels and of the packed weight stream they read.
//
// Network (R/network/nerf.py:49-59): E_p -> 256 x5 -> [E_p+256] -> 256 x3 -> 257 -> [256+E_d] -> 128 -> 3 with
// feat_dim = 256 and any pos_dim E_p <= 64, view_dir_dim E_d <= 32 (struct Net below; the encoded position always
// occupies two 32-wide k-blocks and the direction one, zero-padded -- 63 / 27 for the reference's shipped yaml).
// Other feat_dim / wider inputs run the layer-by-layer kernels of mlp_layered.hip.
//
// The kernels evaluate Y^T = W . X^T with v_mfma_f32_32x32x2_f32: the WEIGHTS are the
// A operand (M dimension = output features) and the ACTIVATIONS are the B operand
// (N dimension = 32 samples of one wavefront).  The C/D fragment of that MFMA holds,
// in lane (j = lane&31, h = lane>>5) and register r of feature block fb, output feature
//      32*fb + (r&3) + 8*(r>>2) + 4*h          of sample j,
// which is exactly the B-operand fragment (k = lane>>5) the NEXT layer needs if its
// 16 MFMA k-steps for that 32-feature block walk r = 0..15.  Activations therefore
// never leave registers between layers; only weights move (L2 -> LDS -> A operand).
//
// Packed stream = [const block][chunk 0][chunk 1]...[chunk 77], consumed linearly, two chunks
// (one "pair" = 64 KiB = 64 k-values) per pipeline step.
// A chunk is the LDS image of W[:, 32 consecutive k] for one layer: 256 rows (output
// features) x 128 B; row n holds eight 16-byte slots, logical slot c = (k%32)/4 is
// stored at physical slot c ^ ((n>>1)&7) so that the ds_read_b128 of a 16-lane group
// (16 distinct rows, same logical slot) touches 16 distinct 16-B slots of a 256-B bank
// row: conflict-free.  The image is what the LDS-DMA copies verbatim (lane-linear).
#pragma once
#include <stdint.h>

namespace mlp {

constexpr int FEAT = 256, HALF = 128;
constexpr int MAX_E_POS = 64, MAX_E_DIR = 32;   // two / one 32-wide k-blocks
constexpr int NUM_LAYERS = 11;

// The network instance a launch works on: NeRF(e_pos, e_dir, 256) and, for the kernels that encode raw points in
// registers, the two PositionalEncoder(3, levels, include_input) in front of it (positional_encoder.py:27-47;
// yaml knobs coord_encode_level / dir_encode_level / include_input, runner_utils.py:584-594).  levels < 0: the
// encoders are not PositionalEncoders -- only pre-encoded inputs are accepted.  Passed to kernels BY VALUE.
struct Net {
    int e_pos, e_dir;          // pos_dim, view_dir_dim
    int l_pos, l_dir;          // encode levels (raw-input kernels)
    int inc_pos, inc_dir;      // include_input
    __host__ __device__ constexpr bool is_default() const {
        return e_pos == 63 && e_dir == 27 && l_pos == 10 && l_dir == 4 && inc_pos == 1 && inc_dir == 1;
    }
    __host__ __device__ constexpr int layer_in(int l) const {
        return l == 0 ? e_pos : l == 5 ? FEAT + e_pos : l == 9 ? FEAT + e_dir : l == 10 ? HALF : FEAT;
    }
    static __host__ __device__ constexpr int layer_out(int l) { return l == 8 ? FEAT + 1 : l == 9 ? HALF : l == 10 ? 3 : FEAT; }
    // flat parameter blob offsets (state_dict order, weight (out,in) then bias)
    __host__ __device__ constexpr int64_t w_offset(int l) const {
        int64_t off = 0;
        for (int i = 0; i < l; ++i) off += (int64_t)layer_out(i) * layer_in(i) + layer_out(i);
        return off;
    }
    __host__ __device__ constexpr int64_t b_offset(int l) const { return w_offset(l) + (int64_t)layer_out(l) * layer_in(l); }
    __host__ __device__ constexpr int64_t param_count() const { return w_offset(NUM_LAYERS); }
};
constexpr Net DEFAULT_NET = {63, 27, 10, 4, 1, 1};
static_assert(DEFAULT_NET.param_count() == 595844, "parameter count of NeRF(63, 27, 256)");

// ---- const block (resident in LDS for the whole kernel), float offsets
constexpr int CB_BIAS = 0;            // 8 x 256 : biases of fc_in, fc_1 .. fc_7
constexpr int CB_BIAS8 = 2048;        // 256     : fc_8.bias[1:257]
constexpr int CB_BIAS9 = 2304;        // 128     : fc_9.bias
constexpr int CB_W8ROW0 = 2432;       // 256     : fc_8.weight[0, :]  (the density row)
constexpr int CB_WOUT = 2688;         // 3 x 128 : fc_out.weight
constexpr int CB_SCALARS = 3072;      // 4       : fc_8.bias[0], fc_out.bias[0..2]
constexpr int CONST_FLOATS = 3328;    // padded to 13 KiB
constexpr int CONST_BYTES = CONST_FLOATS * 4;

// ---- weight chunks
constexpr int CHUNK_ROWS = 256, CHUNK_K = 32;
constexpr int CHUNK_BYTES = CHUNK_ROWS * CHUNK_K * 4;  // 32 KiB
constexpr int CHUNK_FLOATS = CHUNK_BYTES / 4;
// consumption order of the forward stream
constexpr int CH_FC_IN = 0;    // 2 chunks : fc_in, k = encoded position (63 padded to 64)
constexpr int CH_TRUNK1 = 2;   // 4 x 8    : fc_1 .. fc_4
constexpr int CH_FC5_ENC = 34; // 2        : fc_5[:, 0:63]   (skip connection, pos first: nerf.py:108)
constexpr int CH_FC5 = 36;     // 8        : fc_5[:, 63:319]
constexpr int CH_TRUNK6 = 44;  // 2 x 8    : fc_6, fc_7
constexpr int CH_FC8 = 60;     // 8        : fc_8 rows 1..256
constexpr int CH_FC9_DIR = 68; // 1 (+1)   : fc_9[:, 256:256+E_d] (direction, padded to 32), rows 0..127; chunk 69 is zero
                               //            filler so that the stream is a whole number of PAIRS.  The direction goes
                               //            FIRST: its contribution to the accumulators (bias + 27 fma's, in the k order
                               //            of the MFMA chain) depends on the ray only, so the fused render pass computes
                               //            it once per ray, starts fc_9 from that vector and skips this pair -- with
                               //            bit-identical results (render_fused.hip)
constexpr int CH_FC9 = 70;     // 8        : fc_9[:, 0:256]; rows 0..127
constexpr int FWD_CHUNKS = 78;
constexpr int FC9_DIR_PAIR = CH_FC9_DIR / 2;   // the pair a ray-constant direction lets the kernel skip
constexpr int64_t FWD_BYTES = (int64_t)CONST_BYTES + (int64_t)FWD_CHUNKS * CHUNK_BYTES;

// ---- transposed stream for the backward dX chain (chunks of W^T: rows = INPUT feature,
// k = output feature), appended after the forward stream.  Three pairs serve the gradients w.r.t. the network's two
// INPUTS (nerf.py:102, :108, :116: autograd's g_pos, g_view_dir); a dX chain that is not asked for them skips those
// pairs (BWD_INPUT_GRAD_PAIRS -> Pipe::skip_mask).  They are "slot-major": 16 slots of 4 KiB per pair, slot =
// [32 input features x 32 output features] in the row format of a chunk's 32-row block:
//   fc_5[:, :E_p]^T / fc_in^T : slot 2 kb + fb = input block fb (0..1) x output block kb (0..7)
//   fc_9[:, 256:]^T           : slot kb        = the direction block x output block kb (0..3); slots 4..15 zero
constexpr int BW_DIRT = 0;     // 2 (slot-major)  : fc_9[:, 256:256+E_d]^T      -> g_view_dir (FIRST: the accumulators are idle)
constexpr int BW_FC9T = 2;     // 4 chunks : fc_9[:, 0:256]^T     (rows 256 inputs, k = 128 outputs)
constexpr int BW_FC8T = 6;     // 8        : fc_8[1:257, :]^T
constexpr int BW_FC7T = 14;    // 8, then fc_6^T at 22
constexpr int BW_FC5T = 30;    // 8        : fc_5[:, E_p:E_p+256]^T
constexpr int BW_FC4T = 38;    // 8 each: fc_4^T, fc_3^T, fc_2^T, fc_1^T
constexpr int BW_FCINT = 70;   // 2 (slot-major)  : fc_in^T                     -> g_pos (fc_in's share)
constexpr int BW_FC5POST = 72; // 2 (slot-major)  : fc_5[:, 0:E_p]^T            -> g_pos (the skip connection's share; in the
                               //                   epilogue, from the dY5 plane: inside the layer loop the extra block costs
                               //                   hipcc its register allocation -- 52 spills)
constexpr int BWD_CHUNKS = 74;
constexpr unsigned long long BWD_INPUT_GRAD_PAIRS =
    (1ull << (BW_DIRT / 2)) | (1ull << (BW_FC5POST / 2)) | (1ull << (BW_FCINT / 2));
constexpr int64_t BWD_OFFSET = FWD_BYTES;
constexpr int64_t PACKED_BYTES = FWD_BYTES + (int64_t)BWD_CHUNKS * CHUNK_BYTES;

// ring of chunk slots in LDS: 2 pair-slots of 2 chunks
constexpr int RING_SLOTS = 4;
constexpr int PAIR_BYTES = 2 * CHUNK_BYTES;
constexpr int LDS_BYTES = RING_SLOTS * CHUNK_BYTES + CONST_BYTES;  // 144384 <= 160 KiB

// ---- activation record written by the training-mode forward.  MP = M rounded up to a multiple of 128
// rows so that the backward GEMMs may consume whole 32-row tiles; rows >= M hold finite filler here and
// exact zeros in the gradient planes.
// Plane layout "TF" (tile-fragment), shared by the record and by the gradient planes of the dX chain:
// a plane of W features/sample is a sequence of 32-sample tiles of 32 W contiguous floats -- what the dW
// kernel DMAs into LDS in one go -- and inside a tile the data sits in the order the WRITER holds it:
// slot (fb, q) = the 4-feature group {32 fb + 8 q + 4 h + e} of all 32 samples and both lane halves h =
// 64 lanes x 16 B = exactly the 1 KiB one wavefront store instruction writes, so every store is eight full
// 128-byte lines (a row-major plane makes the same instruction 32 scattered 32-byte pieces: the layer
// seams of the dX chain then take 10.4 k cycles instead of 5.4 k).  Inside a slot, lane (i, h) sits at 16-byte
// unit (2 i + h) ^ (2 q): the XOR makes the dW kernel's ds_read_b32 fragments (one sample pair x 32
// consecutive features) hit 32 different LDS banks per half-wave.
//   float planes (offsets in floats, x MP):
//     PL_PE   : encoded position, 64 floats/sample (63 + one zero)
//     PL_H(l) : post-ReLU outputs of fc_in (l=0), fc_1 .. fc_7 (l=7), 256 floats/sample
//     PL_Y8   : fc_8 output rows 1..256 (no ReLU), 256 floats/sample
//     PL_H9   : post-ReLU fc_9 output, 128 floats/sample
//     PL_DE   : encoded direction, 32 floats/sample (27 + zeros)
//   mask planes (after the float planes): 9 planes (h0..h7, h9) of 32 B/sample: for sample m and
//   lane half h, a uint4 at index 2m+h whose dword fb>>1, bit 16*(fb&1)+r is (activation > 0)
//   for the D-fragment register r of feature block fb.
__host__ __device__ constexpr int64_t tf_offset(int width, int64_t m, int k) {  // element (sample m, feature k)
    return (m >> 5) * 32 * width + (((k >> 5) * 4 + ((k >> 3) & 3)) << 8) +
           4 * ((2 * (int)(m & 31) + ((k >> 2) & 1)) ^ (2 * ((k >> 3) & 3))) + (k & 3);
}
constexpr int SAVED_FLOATS_PER_SAMPLE = 64 + 8 * 256 + 256 + 128 + 32;  // 2528
constexpr int SAVED_MASK_PLANES = 9;
constexpr int SAVED_BYTES_PER_SAMPLE = SAVED_FLOATS_PER_SAMPLE * 4 + SAVED_MASK_PLANES * 32;  // 10400
constexpr int TILE_SAMPLES = 128;  // one workgroup pass: 4 wavefronts x 32 samples
__host__ __device__ constexpr int64_t padded_rows(int64_t M) { return (M + TILE_SAMPLES - 1) / TILE_SAMPLES * TILE_SAMPLES; }
__host__ __device__ constexpr int64_t pl_pe(int64_t MP) { return 0; }
__host__ __device__ constexpr int64_t pl_h(int64_t MP, int l) { return MP * (64 + 256 * (int64_t)l); }
__host__ __device__ constexpr int64_t pl_y8(int64_t MP) { return MP * (64 + 256 * 8); }
__host__ __device__ constexpr int64_t pl_h9(int64_t MP) { return MP * (64 + 256 * 9); }
__host__ __device__ constexpr int64_t pl_de(int64_t MP) { return MP * (64 + 256 * 9 + 128); }
__host__ __device__ constexpr int64_t pl_masks(int64_t MP) { return MP * SAVED_FLOATS_PER_SAMPLE; }  // in floats

// ---- gradient planes written by the backward dX chain (workspace, offsets in floats, x MP):
//   DY(l), l = 0..7 : grad w.r.t. the pre-activation of fc_in, fc_1 .. fc_7 (256/sample)
//   DY8             : grad w.r.t. fc_8 output rows 1..256                    (256/sample)
//   DY9             : grad w.r.t. the pre-activation of fc_9                 (128/sample)
//   DSIG            : grad w.r.t. fc_8 output row 0 (density pre-activation) (1/sample)
__host__ __device__ constexpr int64_t dy_plane(int64_t MP, int l) { return MP * 256 * (int64_t)l; }  // l = 0..8
__host__ __device__ constexpr int64_t dy9_plane(int64_t MP) { return MP * 256 * 9; }
__host__ __device__ constexpr int64_t dsig_plane(int64_t MP) { return MP * (256 * 9 + 128); }
//   GY              : grad w.r.t. the three pre-sigmoid colours, [sample][4] (the 4th float is zero) (4/sample)
__host__ __device__ constexpr int64_t gy_plane(int64_t MP) { return MP * (256 * 9 + 128 + 1); }
//   GP, GD          : (input-gradient dX chain only) g_pos (64/sample) and g_view_dir (32/sample), TF layout; a small
//                     kernel writes the callers' row-major (M, E_p) / (M, E_d) tensors from them
__host__ __device__ constexpr int64_t gp_plane(int64_t MP) { return MP * (256 * 9 + 128 + 1 + 4); }
__host__ __device__ constexpr int64_t gd_plane(int64_t MP) { return MP * (256 * 9 + 128 + 1 + 4 + 64); }
constexpr int DY_FLOATS_PER_SAMPLE = 256 * 9 + 128 + 1 + 4 + 96;

// ---- bf16 inference stream (BASELINE configs[2]: bf16 weights on v_mfma_f32_32x32x16_bf16).
// [const block (fp32, as above)][sub-step 0]...[sub-step 36]; a sub-step = 32 KiB = what one pipeline step of
// mlp_forward_bf16.hip moves into one LDS ring slot.  A full chunk = the LDS image of W[:, 32 k-values] in bf16:
// 256 rows x 64 B.  Row n holds four 16-byte A fragments (8 bf16 each), one per (k-step s, lane half h): element e
// is W[n][32 kb + 16 s + 8 (e>>2) + 4 h + (e&3)], i.e. exactly the features the D fragment of the previous layer
// holds in registers 8s..8s+7 of lane half h.  Fragment (s,h) of row n sits at slot (2s+h) ^ ((n>>2)&3):
// conflict-free ds_read_b128.  A sub-step of a 256-row layer is two chunks (k-blocks 2j, 2j+1).  fc_9 has 128
// rows: its chunks are 128 rows x 64 B = 8 KiB (same row format) and three of them share a sub-step.
//   sub 0         fc_in   (encoded position, 63 -> 64)          subs 22..29   fc_6, fc_7
//   subs 1..16    fc_1 .. fc_4  (4 per layer)                   subs 30..33   fc_8 rows 1..256
//   sub 17        fc_5[:, 0:63]   (skip connection, pos first)  subs 34,35    fc_9[:, 0:96], fc_9[:, 96:192]
//   subs 18..21   fc_5[:, 63:319]                               sub 36        fc_9[:, 192:256], fc_9[:, 256:283] (dir)
constexpr int B16_CHUNK_BYTES = CHUNK_ROWS * CHUNK_K * 2;   // 16 KiB
constexpr int B16_HALF_CHUNK_BYTES = B16_CHUNK_BYTES / 2;   // 8 KiB: 128 rows (fc_9)
constexpr int B16_SUB_BYTES = 2 * B16_CHUNK_BYTES;          // 32 KiB
constexpr int B16_SUBS = 37;
constexpr int64_t B16_PACKED_BYTES = (int64_t)CONST_BYTES + (int64_t)B16_SUBS * B16_SUB_BYTES;
__host__ __device__ constexpr int b16_frag_offset(int n, int slot) { return n * 64 + ((slot ^ ((n >> 2) & 3)) << 4); }

// ---- split-f16 inference stream ("f16x2": fp32-grade results from the f16 matrix pipe, mlp_forward_f16x2.hip).
// Every weight is scaled by its layer's power of two 2^s (max |W_l| -> [2^13, 2^14): the LOW part of a weight then stays
// a normal f16 for every weight above 2^-16 of the largest) and split in two f16 parts, hi = f16(w), lo = f16(w - hi);
// activations are split the same way in the layer seams and every k-step forms lo.hi + hi.lo + hi.hi on
// v_mfma_f32_16x16x32_f16 (fp32 accumulate): 22 significand bits per operand, exact part products.
// [const block (fp32)][sub-step 0] ... [sub-step 72]; a sub-step = 32 KiB = one LDS ring slot:
//   256-row layers: ONE 32-wide k-block, [hi image 16 KiB][lo image 16 KiB]; an image = 256 rows x 64 B, row n holds four
//     16-byte A fragments (8 f16), one per lane group g = lane >> 4: element e is W[n][32 kb + 16 (e>>2) + 4 g + (e&3)] --
//     the features the D fragments of output blocks 2 kb and 2 kb + 1 (16 features each) of the previous layer hold in
//     lane group g.  Fragment g of row n sits at slot g ^ sigma((n>>2)&3), sigma = (0, 2, 3, 1) (f2_frag_offset):
//     ds_read_b128 serves the lanes {0-3, 12-15, 20-23, 24-27} | {4-7, 8-11, 16-19, 28-31} | the same + 32 together, and
//     with lane = 16 g + (row & 15) the bf16 stream's sigma = identity puts two row quads of such a set on the same
//     banks (PMC: 48 % of the LDS-active cycles were bank conflicts, profiles/r06_pmc_f16x2_first_cut.txt);
//     scripts/lds_b128_probe.hip times all 24 sigma.
//   fc_9 (128 rows): images of 8 KiB, TWO k-blocks per sub-step: [hi kb][lo kb][hi kb+1][lo kb+1]
//   (the shipped 63 / 27: two position k-blocks, one direction k-block; F2Layout below for the wider inputs)
//   subs 0,1      fc_in  (encoded position, two k-blocks)         subs 44..59   fc_6, fc_7
//   subs 2..33    fc_1 .. fc_4  (8 per layer)                     subs 60..67   fc_8 rows 1..256
//   subs 34,35    fc_5[:, 0:E_p]  (skip connection, pos first)    subs 68..71   fc_9[:, 0:256], two k-blocks each
//   subs 36..43   fc_5[:, E_p:E_p+256]                            sub 72        fc_9[:, 256:256+E_d] (direction) + zeros
// const block: the fp32 layout above with the biases PRE-SCALED by their layer's 2^s (the C fragment starts as the bias),
// plus the ten factors 2^-s (layers fc_in .. fc_9) the seams multiply the accumulators with, and 2^s for the packer.
constexpr int F2_IMAGE_BYTES = CHUNK_ROWS * CHUNK_K * 2;    // 16 KiB
constexpr int F2_SUB_BYTES = 2 * F2_IMAGE_BYTES;            // 32 KiB
// The split kernel also serves the wider inputs the yaml can name (coord_encode_level 11..20 -> pos_dim <= 123,
// dir_encode_level 5..10 -> view_dir_dim <= 63, runner_utils.py:584-612): NPOS = 2 | 3 | 4 position k-blocks (at least the
// fused family's two) and NDIR = 1 | 2 direction k-blocks, with feat_dim 256.  Sub-steps in consumption order:
//   fc_in NPOS | fc_1..fc_4 32 | fc_5 position NPOS | fc_5 8 | fc_6, fc_7 16 | fc_8 8 | fc_9 ceil((8 + NDIR) / 2)
constexpr int F2_MAX_E_POS = 128, F2_MAX_E_DIR = 64;
struct F2Layout {
    int npos, ndir;
    __host__ __device__ constexpr int sub_fc1() const { return npos; }
    __host__ __device__ constexpr int sub_fc5_pos() const { return npos + 32; }
    __host__ __device__ constexpr int sub_fc5() const { return 2 * npos + 32; }
    __host__ __device__ constexpr int sub_fc6() const { return 2 * npos + 40; }
    __host__ __device__ constexpr int sub_fc8() const { return 2 * npos + 56; }
    __host__ __device__ constexpr int sub_fc9() const { return 2 * npos + 64; }
    __host__ __device__ constexpr int subs() const { return 2 * npos + 64 + (8 + ndir + 1) / 2; }
    // the fused family's stream (npos 2, ndir 1) is followed by the TRANSPOSED stream of the reverse chain
    // (mlp_bwd_dx_f16x2_kernel): fc_9[:, 0:256]^T in 4 sub-steps, then fc_8[1:257]^T, fc_7^T .. fc_1^T (fc_5: its 256
    // feature columns) in 8 each -- image row = INPUT feature of the layer, k = its output features, same fragment format
    __host__ __device__ constexpr int bwd_subs() const { return (npos == 2 && ndir == 1) ? 4 + 8 * 8 : 0; }
    __host__ __device__ constexpr int64_t packed_bytes() const { return (int64_t)CONST_BYTES + (int64_t)(subs() + bwd_subs()) * F2_SUB_BYTES; }
};
__host__ __device__ constexpr F2Layout f2_layout(int e_pos, int e_dir) {
    return F2Layout{e_pos <= 64 ? 2 : (e_pos + 31) / 32, (e_dir + 31) / 32};
}
constexpr int F2_SUBS = f2_layout(63, 27).subs();           // 73: the shipped network
static_assert(F2_SUBS == 73, "split-f16 stream of NeRF(63, 27, 256)");
constexpr int F2_BWD_SUBS = f2_layout(63, 27).bwd_subs();   // 68
constexpr int F2_CB_UNSCALE = CB_SCALARS + 4;               // 10 floats: 2^-s of fc_in .. fc_9
constexpr int F2_CB_SCALE = F2_CB_UNSCALE + 10;             // 10 floats: 2^s
static_assert(F2_CB_SCALE + 10 <= CONST_FLOATS, "const block");
// (the packer's 10 x 16 partial maxima follow, mlp_pack.hip) ... and the last 16 words are, in the LDS copy of the reverse
// chain, the workgroup's running maxima of |dY| per gradient plane (mlp_forward_f16x2.hip: ds_max_u32, flushed at the end)
constexpr int F2_CB_PLANE_MAX = CONST_FLOATS - 16;
constexpr int64_t F2_PACKED_BYTES = f2_layout(63, 27).packed_bytes();
__host__ __device__ constexpr int f2_sigma(int q) { return (0x78 >> (2 * q)) & 3; }     // 0, 2, 3, 1
__host__ __device__ constexpr int f2_frag_offset(int n, int g) { return n * 64 + ((g ^ f2_sigma((n >> 2) & 3)) << 4); }

// physical byte offset, inside a chunk image, of the 16-byte slot holding
// W[row n][k-group c] (c = (k % 32) / 4)
__host__ __device__ constexpr int chunk_slot_offset(int n, int c) { return n * 128 + ((c ^ ((n >> 1) & 7)) << 4); }

}  // namespace mlp
