// nerf_layout.h -- index math shared by the weight packer, the fused MLP
// kernels and the host-side layout self-test.  Plain C++ (no HIP types), all
// functions constexpr so host and device compile the same formulas.
//
// The network (reference utils/nets.py:16-32) is re-expressed as 11 internal
// layers whose OUTPUT features sit on MFMA rows and whose points sit on MFMA
// columns (H^T = W . X^T), so that an accumulator tile of one layer is directly
// the B operand of the next (cdna_hip_programming.md section 3, "An accumulator
// tile as the next MFMA's operand"):
//
//   L0      posx(63->64 slots)            -> 256  relu   layers_0.0
//   L1..L4  256                           -> 256  relu   layers_0.{2,4,6,8}
//   L5      256 (h5) + posx 64 slots      -> 256  relu   skip_conn_layer.0  [h ; x]
//   L6,L7   256                           -> 256  relu   layers_1.{0,2}
//   L8      256 (h8)                      -> 288  none   rows 0..255 layers_2, row 256 sigma_fc, rest 0
//   L9      256 (h9) + posd 32 slots      -> 128  relu   color_fc.0        [h ; d]
//   L10     128                           -> 32   none   rows 0..2 color_fc.2, rest 0
#pragma once

#if defined(__HIPCC__)
#define NL_HD __host__ __device__
#else
#define NL_HD
#endif

namespace nerf_layout {

constexpr int NUM_LAYERS = 11;
constexpr int PARAM_COUNT = 595844;
// largest samples-per-ray the fused render kernel composites in its LDS ring (mlp_bf16_16.hip);
// longer rays take the two-launch path (MLP -> raw/ts in HBM -> composite.hip)
constexpr int FUSED_RENDER_MAX_N = 768;

// offsets of the 24 tensors inside the flat fp32 parameter vector
// (PARAM_SPECS order of utils/synthetic.py == state_dict order of the reference)
constexpr int OFF_L0_W = 0;                         // 256x63
constexpr int OFF_L0_B = OFF_L0_W + 256 * 63;
constexpr int OFF_L1_W = OFF_L0_B + 256;            // 4 x (256x256 + 256)
constexpr int OFF_SKIP_W = OFF_L1_W + 4 * (256 * 256 + 256);   // 256x319
constexpr int OFF_SKIP_B = OFF_SKIP_W + 256 * 319;
constexpr int OFF_L6_W = OFF_SKIP_B + 256;          // 2 x (256x256 + 256)
constexpr int OFF_SIG_W = OFF_L6_W + 2 * (256 * 256 + 256);    // 1x256
constexpr int OFF_SIG_B = OFF_SIG_W + 256;
constexpr int OFF_L2_W = OFF_SIG_B + 1;             // 256x256
constexpr int OFF_L2_B = OFF_L2_W + 256 * 256;
constexpr int OFF_C0_W = OFF_L2_B + 256;            // 128x283
constexpr int OFF_C0_B = OFF_C0_W + 128 * 283;
constexpr int OFF_C1_W = OFF_C0_B + 128;            // 3x128
constexpr int OFF_C1_B = OFF_C1_W + 3 * 128;
static_assert(OFF_C1_B + 3 == PARAM_COUNT, "parameter offsets");
// Data-parallel training reduces the flat gradient vector in two pieces: [0, SPLIT) = layers_0.* and
// [SPLIT, PARAM_COUNT) = skip_conn_layer ... color_fc.2; the second piece is computed first (dw_gemm.hip).
constexpr int GRAD_BUCKET_SPLIT = OFF_SKIP_W;          // 279,552 floats (1.12 MB) | 316,292 floats (1.27 MB)
static_assert(GRAD_BUCKET_SPLIT % 4 == 0, "16-byte aligned buckets");

struct LayerDesc {
    int w_off, b_off;     // into the flat parameter vector
    int ld;               // row stride (= true K) of the source weight
    int rows;             // true output rows held by the source weight
    int chain_k;          // input features taken from the previous layer's accumulators
    int extra_slots;      // posx (64) / posd (32) slots appended after the chain part
    int extra_col0;       // first source column of the appended part
    int extra_kind;       // 0 none, 1 posx, 2 posd
    int mt;               // output tiles of 32 rows
    int relu;
};

NL_HD constexpr LayerDesc layer_desc(int L) {
    return L == 0 ? LayerDesc{OFF_L0_W, OFF_L0_B, 63, 256, 0, 64, 0, 1, 8, 1}
         : L <= 4 ? LayerDesc{OFF_L1_W + (L - 1) * (256 * 256 + 256),
                              OFF_L1_W + (L - 1) * (256 * 256 + 256) + 256 * 256,
                              256, 256, 256, 0, 0, 0, 8, 1}
         : L == 5 ? LayerDesc{OFF_SKIP_W, OFF_SKIP_B, 319, 256, 256, 64, 256, 1, 8, 1}
         : L <= 7 ? LayerDesc{OFF_L6_W + (L - 6) * (256 * 256 + 256),
                              OFF_L6_W + (L - 6) * (256 * 256 + 256) + 256 * 256,
                              256, 256, 256, 0, 0, 0, 8, 1}
         : L == 8 ? LayerDesc{OFF_L2_W, OFF_L2_B, 256, 256, 256, 0, 0, 0, 9, 0}
         : L == 9 ? LayerDesc{OFF_C0_W, OFF_C0_B, 283, 128, 256, 32, 256, 2, 4, 1}
         :          LayerDesc{OFF_C1_W, OFF_C1_B, 128, 3, 128, 0, 0, 0, 1, 0};
}

// padded K of a layer as the kernels see it
NL_HD constexpr int layer_k(int L) { return layer_desc(L).chain_k + layer_desc(L).extra_slots; }

// ---- k-permutations -------------------------------------------------------
// f32, mfma_f32_16x16x4f32 (16-row tiles): k-step s covers 4 k's, one per lane
// group g = lane>>4.  Accumulator register i of 16-row tile t holds row 4g + i,
// so that register IS k-step 4t + i of the next layer:
NL_HD constexpr int chain_feat_f32(int s, int g) {
    return 16 * (s >> 2) + 4 * g + (s & 3);
}

// posx / posd slots (4 lane groups, shared by the f32 and the 16-bit kernels):
// group g evaluates, per coordinate, the five
// (level, trig) pairs idx = 5g..5g+4 of the 20 (idx = 2*level + trig), and one
// raw coordinate.  slot t in [0,16).
NL_HD constexpr int posx_col_f32(int t, int g) {
    return t < 15 ? 3 + 20 * (t / 5) + 5 * g + t % 5
         :          (g < 3 ? g : -1);
}
// posd, slot t in [0,8): per coordinate the pair (level g, trig j)
NL_HD constexpr int posd_col_f32(int t, int g) {
    return t < 6 ? 3 + 8 * (t / 2) + 2 * g + t % 2
         : t == 6 ? (g < 3 ? g : -1)
         :          -1;
}

// source column of layer L's weight for MFMA k position,
// f32 form: ks = k-step (4 k's each), g = lane group
NL_HD constexpr int src_col_f32(int L, int ks, int g) {
    const LayerDesc d = layer_desc(L);
    const int chain_ks = d.chain_k / 4;
    if (ks < chain_ks) return chain_feat_f32(ks, g);
    const int t = ks - chain_ks;
    const int c = d.extra_kind == 1 ? posx_col_f32(t, g) : posd_col_f32(t, g);
    return c < 0 ? -1 : d.extra_col0 + c;
}

// weight value W_L[row][col] (row < mt*32) out of the flat parameter vector,
// 0 for padding; L8 row 256 is sigma_fc.
template <class T>
NL_HD inline float weight_at(const T* params, int L, int row, int col) {
    if (col < 0) return 0.f;
    const LayerDesc d = layer_desc(L);
    if (L == 8 && row == 256) return params[OFF_SIG_W + col];
    if (row >= d.rows) return 0.f;
    return params[d.w_off + row * d.ld + col];
}
template <class T>
NL_HD inline float bias_at(const T* params, int L, int row) {
    const LayerDesc d = layer_desc(L);
    if (L == 8 && row == 256) return params[OFF_SIG_B];
    if (row >= d.rows) return 0.f;
    return params[d.b_off + row];
}

// ---- packed 16-bit image, 16-row tiles (v_mfma_f32_16x16x32_bf16 / _f16) ----------
// A wave's 64 lanes are 16 points x 4 lane groups g; a k-step covers 32 k's,
// lane group g holding elements j = 0..7 (hardware k = 8g + j).  Two stacked
// 16-row accumulator tiles (2q, 2q+1) give lane (c,g) rows 4g..4g+3 of each:
// converted to bf16 they ARE the B fragment of the next layer's k-step q with
NL_HD constexpr int chain_feat_b16(int q, int g, int j) {
    return 32 * q + 16 * (j >> 2) + 4 * g + (j & 3);
}
// the extra (posx / posd) slots reuse the 4-lane-group maps of the f32 form
NL_HD constexpr int src_col_b16(int L, int ks, int g, int j) {
    const LayerDesc d = layer_desc(L);
    const int chain_ks = d.chain_k / 32;
    if (ks < chain_ks) return chain_feat_b16(ks, g, j);
    const int t = 8 * (ks - chain_ks) + j;
    const int c = d.extra_kind == 1 ? posx_col_f32(t, g) : posd_col_f32(t, g);
    return c < 0 ? -1 : d.extra_col0 + c;
}
NL_HD constexpr int b16_mt(int L) { return L == 8 ? 17 : L == 10 ? 1 : layer_desc(L).mt * 2; }
NL_HD constexpr int b16_ks(int L) { return layer_k(L) / 32; }       // 1 KiB fragments per 16-row tile
NL_HD constexpr int b16_layer_off_kib(int L) {
    int o = 0;
    for (int i = 0; i < L; ++i) o += b16_mt(i) * b16_ks(i);
    return o;
}
constexpr int B16_WEIGHT_KIB = b16_layer_off_kib(NUM_LAYERS);        // 1172
NL_HD constexpr int b16_bias_off(int L) {
    int o = 0;
    for (int i = 0; i < L; ++i) o += b16_mt(i) * 16;
    return o;
}
constexpr int B16_BIAS_FLOATS = b16_bias_off(NUM_LAYERS);            // 2464
// the packed 16-bit buffer (bf16 or fp16): [weight image | bias table (fp32, natural row order) | status block]
// Status block: 256 bytes of sticky uint32 flags (0 / 1), zeroed by the packer.  Word 0 is set by the MLP kernels when
// a point's output is not finite, word 1 by the packer when a weight does not fit the operand type (|w| > 65504 in
// fp16).  One word per writer: each flag is a plain store of 1.
constexpr long long B16_STATUS_OFF = (long long)B16_WEIGHT_KIB * 1024 + B16_BIAS_FLOATS * 4;
constexpr int B16_STATUS_BYTES = 256;
constexpr int NERF_STATUS_WORD_NONFINITE = 0, NERF_STATUS_WORD_WEIGHT_RANGE = 1;
constexpr long long B16_IMAGE_BYTES = B16_STATUS_OFF + B16_STATUS_BYTES;
static_assert(B16_STATUS_OFF % 16 == 0, "aligned status block");

// ---- backward (dX chain) image, bf16, 16-row tiles -------------------------------
// Training backward of the dense layers: dX = W^T dY.  The same on-chip chaining
// as the forward with the roles transposed: rows of an MFMA tile are INPUT
// features of forward layer `wl`, the k dimension runs over its OUTPUT
// features (the accumulators of the previous backward layer), columns are
// points.  10 backward layers, in execution order:
//   b0  C1^T        : k = drgb (3, one custom k-step)     -> d c      (128 rows, masked by c  > 0)
//   b1  C0^T[:256]  : k = d c (128)                       -> d h9     (256 rows, no mask: L8 is linear)
//   b2  [L2;sigma]^T: k = d h9 (256) + dsigma (1 k-step)  -> d h8     (masked by h8 > 0)
//   b3..b4 L7^T,L6^T                                      -> d h7, d h6
//   b5  L5^T[:256]  (skip layer, h part)                  -> d h5
//   b6..b9 L4^T..L1^T                                     -> d h4 .. d h1
// (L0^T is never needed: the inputs carry no gradient.)
constexpr int NUM_BWD = 10;
struct BwdDesc {
    int wl;        // forward layer whose weight is transposed
    int rows;      // rows of dX = input features taken from that layer (h part only)
    int chain_k;   // k taken from the previous backward layer's accumulators
    int extra;     // 1: one more k-step fed from d_raw (b0: drgb, b2: dsigma)
    int mask_act;  // saved forward activation (internal layer index) gating dX, -1 = none
};
NL_HD constexpr BwdDesc bwd_desc(int b) {
    return b == 0 ? BwdDesc{10, 128, 0, 1, 9}
         : b == 1 ? BwdDesc{9, 256, 128, 0, -1}
         : b == 2 ? BwdDesc{8, 256, 256, 1, 7}
         :          BwdDesc{10 - b, 256, 256, 0, 9 - b};      // b3: wl 7, mask act 6 ... b9: wl 1, mask act 0
}
NL_HD constexpr int bwd_ks(int b) { return bwd_desc(b).chain_k / 32 + bwd_desc(b).extra; }
NL_HD constexpr int bwd_mt(int b) { return bwd_desc(b).rows / 16; }
// forward OUTPUT row (of layer wl) that sits at k position (ks, g, j); -1 = padding
NL_HD constexpr int bwd_src_out(int b, int ks, int g, int j) {
    const BwdDesc d = bwd_desc(b);
    if (ks < d.chain_k / 32) return chain_feat_b16(ks, g, j);
    // custom k-step: lane group 0 carries drgb (b0: rows 0..2) or dsigma (b2: row 256)
    return b == 0 ? ((g == 0 && j < 3) ? j : -1) : ((g == 0 && j == 0) ? 256 : -1);
}
NL_HD constexpr int bwd_layer_off_kib(int b) {
    int o = 0;
    for (int i = 0; i < b; ++i) o += bwd_mt(i) * bwd_ks(i);
    return o;
}
constexpr int BWD_WEIGHT_KIB = bwd_layer_off_kib(NUM_BWD);           // 1112
constexpr long long BWD_IMAGE_BYTES = (long long)BWD_WEIGHT_KIB * 1024;

// saved forward activations (bf16), index = internal layer: L0..L7 post-ReLU (256 features),
// L8 = h9 linear (256), L9 = c post-ReLU (128).  Layout "point-blocked": per layer, per tile of
// 256 points, a 128 KiB block [feature chunk c = f/8 (32)][point in tile (256)][8 bf16 = 16 B].
// The MLP kernels hold a point per lane, so the 16 lanes of a quarter-wave write 16 consecutive
// 16-byte granules of one chunk: 256 contiguous bytes per quarter-wave instead of 16 rows (the
// row-major form cost 64 separate L2 write requests per store instruction and made the training
// kernels store-issue-bound).  The dW kernel (dw_gemm.hip) reads 32-point slabs of a block with
// LDS-DMA, 256 contiguous bytes per quarter-wave as well.  Granules of points >= P are never
// written nor read.
NL_HD constexpr int act_width(int L) { return L == 9 ? 128 : 256; }
constexpr int ACT_TILE_PTS = 256;
constexpr long long ACT_BLOCK_BYTES = 256 * 512;                        // one (layer, tile) block
NL_HD constexpr long long act_tiles(long long P) { return (P + ACT_TILE_PTS - 1) / ACT_TILE_PTS; }
NL_HD constexpr long long act_layer_stride(long long P) { return act_tiles(P) * ACT_BLOCK_BYTES; }
NL_HD constexpr long long act_offset_bytes(int L, long long P) { return (long long)L * act_layer_stride(P); }
// byte offset of feature f of point p inside layer L
NL_HD constexpr long long act_elem_offset(int L, long long p, int f, long long P) {
    return act_offset_bytes(L, P) + (p / ACT_TILE_PTS) * ACT_BLOCK_BYTES
         + ((long long)(f / 8) * ACT_TILE_PTS + p % ACT_TILE_PTS) * 16 + (f % 8) * 2;
}
NL_HD constexpr long long acts_bf16_bytes(long long P) { return 10 * act_layer_stride(P); }
// saved backward pre-activation gradients dY (bf16): same blocks, dY[L] has the width of layer L's output

// ReLU masks of the saved activations, one bit per feature, in the MLP kernels' own register
// layout so the forward writes and the backward reads them as coalesced dwords: per (layer L,
// tile of 256 points) 4 dwords x 512 threads.  Thread (wave, lane) holds, for its column block
// cb and feature-pair group Q>>2, dword cb*2 + (Q>>2): bit (Q&3)*4 + j <-> the low bf16 of
// word j of fragment Q, bit 16 + (Q&3)*4 + j the high one (word j = features 32Q + 16(j>>1) +
// 4g + 2(j&1) + {0,1} of the lane's point).  They follow the bf16 activations in the buffer.
constexpr int MASK_TILE_PTS = 256;
NL_HD constexpr long long mask_tiles(long long P) { return (P + MASK_TILE_PTS - 1) / MASK_TILE_PTS; }
NL_HD constexpr long long mask_region_offset(long long P) { return acts_bf16_bytes(P); }
NL_HD constexpr long long mask_offset_bytes(int L, long long tile, int dword, long long P) {
    return mask_region_offset(P) + (((long long)L * mask_tiles(P) + tile) * 4 + dword) * 2048;   // + tid * 4
}
NL_HD constexpr long long acts_total_bytes(long long P) { return acts_bf16_bytes(P) + 10 * mask_tiles(P) * 8192; }

// ---- 8-bit storage form of the same two buffers (training step, storage = e4m3) ------------------------------------------
// What the forward saves for dW and what the dX chain writes for it can be stored as OCP e4m3 with one power-of-two
// exponent per (32 features x 32 points): dW_l = dY_l^T X_l and db_l are sums over ~10^5..10^6 points, the only readers,
// and the rounding of their operands stays inside the gradient criterion (the chain itself keeps
// bf16 in registers).  Half the bytes of the step's three HBM-bound kernels.
//   data: per layer, per tile of 256 points a 64 KiB block [feature chunk c = f/16 (16)][point in tile (256)][16 x e4m3];
//         the 32 points of a wave write 512 contiguous bytes per chunk, the dW kernel reads 64-point runs of 1 KiB;
//   exponents: per layer, per 32-point block (one MLP wave) 8 bytes, byte Q = biased power-of-two exponent s (e8m0: the
//         stored byte b means value = e4m3(b) * 2^(s - 127)) of features 32Q .. 32Q+31 -- the block scale of
//         v_mfma_scale_f32_32x32x64_f8f6f4.  The MLP kernels choose ONE exponent per group of four such fragments (128
//         features x 32 points; bytes 4k .. 4k+3 are equal) so that the group's largest magnitude lands in [128, 256]: the
//         wave-wide maximum is paid once per group (nerf_device.h store_group_f8);
//   (acts only) the ReLU masks, unchanged, behind them.
// Narrow operands of the dW products (encoder rows, the packed d_raw) use the same form with their own chunk count.
constexpr long long F8_BLOCK_BYTES = 256 * 256;
constexpr int F8_SCALE_BYTES_PER_BLOCK = 8;
NL_HD constexpr long long f8_layer_stride(long long P) { return act_tiles(P) * F8_BLOCK_BYTES; }
NL_HD constexpr long long f8_offset_bytes(int L, long long P) { return (long long)L * f8_layer_stride(P); }
NL_HD constexpr long long f8_elem_offset(int L, long long p, int f, long long P) {
    return f8_offset_bytes(L, P) + (p / ACT_TILE_PTS) * F8_BLOCK_BYTES + ((long long)(f / 16) * ACT_TILE_PTS + p % ACT_TILE_PTS) * 16 + f % 16;
}
NL_HD constexpr long long f8_scale_layer_stride(long long P) { return act_tiles(P) * (ACT_TILE_PTS / 32) * F8_SCALE_BYTES_PER_BLOCK; }
NL_HD constexpr long long f8_scale_offset_bytes(int L, long long P) { return 10 * f8_layer_stride(P) + (long long)L * f8_scale_layer_stride(P); }
NL_HD constexpr long long f8_data_bytes(long long P) { return 10 * f8_layer_stride(P) + 10 * f8_scale_layer_stride(P); }   // dY buffer
NL_HD constexpr long long f8_mask_region_offset(long long P) { return (f8_data_bytes(P) + 255) / 256 * 256; }
NL_HD constexpr long long f8_mask_offset_bytes(int L, long long tile, int dword, long long P) {
    return f8_mask_region_offset(P) + (((long long)L * mask_tiles(P) + tile) * 4 + dword) * 2048;
}
NL_HD constexpr long long f8_acts_total_bytes(long long P) { return f8_mask_region_offset(P) + 10 * mask_tiles(P) * 8192; }
// a narrow operand of W features (W % 16 == 0): [tile][chunk (W/16)][point (256)][16 B], then its exponents [block][8]
NL_HD constexpr long long f8_narrow_block_bytes(int W) { return (long long)(W / 16) * ACT_TILE_PTS * 16; }
NL_HD constexpr long long f8_narrow_scale_offset(int W, long long P) { return act_tiles(P) * f8_narrow_block_bytes(W); }
NL_HD constexpr long long f8_narrow_bytes(int W, long long P) { return f8_narrow_scale_offset(W, P) + f8_scale_layer_stride(P); }

// ---- packed f32 image -----------------------------------------------------
// 16-row output tiles (mfma_f32_16x16x4f32).  chunk = (layer, t): K/4 k-steps
// x 64 lanes x 4 B, stored [ks/4][lane][4] so one ds_read_b128 per lane covers
// 4 consecutive k-steps.  A chunk is K*64 bytes, padded to a multiple of 4 KiB so
// the 4 waves of a workgroup each move the same number of 1 KiB pieces.
NL_HD constexpr int f32_mt(int L) { return L == 8 ? 17 : L == 10 ? 1 : layer_desc(L).mt * 2; }
NL_HD constexpr int f32_ks(int L) { return layer_k(L) / 4; }
NL_HD constexpr int f32_chunk_kib(int L) { return (layer_k(L) / 16 + 3) / 4 * 4; }
NL_HD constexpr int f32_layer_off_kib(int L) {
    int o = 0;
    for (int i = 0; i < L; ++i) o += f32_mt(i) * f32_chunk_kib(i);
    return o;
}
constexpr int F32_WEIGHT_KIB = f32_layer_off_kib(NUM_LAYERS);
NL_HD constexpr int f32_bias_off(int L) {
    int o = 0;
    for (int i = 0; i < L; ++i) o += f32_mt(i) * 16;
    return o;
}
constexpr int F32_BIAS_FLOATS = f32_bias_off(NUM_LAYERS);
constexpr long long F32_PACKED_BYTES = (long long)F32_WEIGHT_KIB * 1024 + F32_BIAS_FLOATS * 4;
NL_HD constexpr int f32_num_chunks() {
    int c = 0;
    for (int i = 0; i < NUM_LAYERS; ++i) c += f32_mt(i);
    return c;
}
constexpr int F32_NUM_CHUNKS = f32_num_chunks();                     // 154

}  // namespace nerf_layout
