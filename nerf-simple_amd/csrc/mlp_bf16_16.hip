// mlp_bf16_16.hip -- the fused 16-bit MLP: sampling + positional encoding + 12 dense layers in
// one launch, on v_mfma_f32_16x16x32_bf16 (and, built with -DNERF_HALF, on ..._f16).
// Replaces reference utils/rendering.py:24-40 + utils/xyz.py:6-36 + utils/nets.py:34-43.
//
// H^T = W . X^T: output features on MFMA rows, points on MFMA columns / lanes (nerf_layout.h).
//   * a wave owns 32 points = two 16-point column blocks; 64 lanes = 16 points x 4 lane groups;
//   * two stacked 16-row accumulator tiles (2q, 2q+1), bias-initialised, ReLU'd and converted
//     pairwise, ARE the B fragment of the next layer's k-step q (32 features): activations never
//     leave the registers;
//   * weights stream L2 -> LDS by LDS-DMA in chunks of four 16-row tiles (eight for the K = 64 first
//     layer; 38 chunks per tile), double buffered, one barrier per chunk; one weight fragment read
//     from LDS (16 rows x 32 k, 1 KiB) feeds two MFMAs (the two column blocks);
//   * the chunk barrier sits three fragments before the END of a chunk and the next chunk's first
//     fragments are requested right behind it, so no chunk starts with an LDS round trip; the next
//     chunk's DMA pieces go out one per four MFMAs (chunk_step);
//   * the sigma head rides as row 256 of the layers_2 product, the rgb head is one 16-row tile;
//   * a workgroup = 8 waves = a 256-point tile, persistent over tiles;
//   * COMP (the render path): compositing (utils/rendering.py:47-85) runs in the same launch.  A
//     workgroup owns a contiguous range of RAYS; each tile drops its 256 x (rgb, sigma, t) into an
//     LDS ring of 1024 samples, and whenever 8 rays are complete (or the ring is full) every wave
//     composites one ray with the routine composite.hip uses (composite_device.h): bit-identical
//     pixels, and raw[B,N,4] / ts[B,N] (20 B per sample) never go to HBM.
// The chip is power/DVFS-limited on this kernel and holds a higher clock on the 16x16x32 shape
// than on 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7).
#include "composite_device.h"
#include <utility>

using namespace nerf_layout;

// The same source builds the bf16 kernel (default) and, with -DNERF_HALF, the
// fp16 one (v_mfma_f32_16x16x32_f16: same cycles, 11-bit mantissa instead of 8).
#ifdef NERF_HALF
typedef _Float16 elem_t;
#define NERF_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#define NERF_KERNEL nerf_mlp_f16_16_kernel
#define NERF_LAUNCH nerf_amd_launch_mlp_f16_16
#else
typedef __bf16 elem_t;
#define NERF_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define NERF_KERNEL nerf_mlp_bf16_16_kernel
#define NERF_LAUNCH nerf_amd_launch_mlp_bf16_16
#endif
typedef elem_t ex8 __attribute__((ext_vector_type(8)));
typedef elem_t ex2 __attribute__((ext_vector_type(2)));

namespace {

// NCB 16-point column blocks per wave: 8 waves (2 per SIMD) x 2 blocks.  (4 waves x 4 blocks with
// the accumulators in AGPRs halves the LDS weight reads but measured 4.5 % slower: DESIGN.md section 5.)
constexpr int NCB = 2;
constexpr int WAVES = 16 / NCB;
constexpr int TILE_PTS = WAVES * 16 * NCB;
// 16-row output tiles per weight chunk: 4 (64 rows, 8..40 KiB), and 8 for the K = 64 first layer, whose
// tiles are two fragments each (one barrier per 16 MFMAs otherwise).  The chunk count stays even: the
// double buffer's parity is cyclic over tiles.
__host__ __device__ constexpr int tpc(int L) { return L == 0 ? 8 : 4; }

__host__ __device__ constexpr int layer_chunks(int L) { return (b16_mt(L) + tpc(L) - 1) / tpc(L); }
__host__ __device__ constexpr int chunk_first(int L) {
    int c = 0;
    for (int i = 0; i < L; ++i) c += layer_chunks(i);
    return c;
}
constexpr int NUM_CHUNKS = chunk_first(NUM_LAYERS);           // 38
__host__ __device__ constexpr int chunk_layer(int cc) {
    int L = 0;
    while (cc >= layer_chunks(L)) { cc -= layer_chunks(L); ++L; }
    return L;
}
__host__ __device__ constexpr int chunk_tiles(int cc) {
    const int L = chunk_layer(cc), C = cc - chunk_first(L);
    const int left = b16_mt(L) - C * tpc(L);
    return left < tpc(L) ? left : tpc(L);
}
__host__ __device__ constexpr int chunk_kib(int cc) { return chunk_tiles(cc) * b16_ks(chunk_layer(cc)); }
__host__ __device__ constexpr int chunk_off_kib(int cc) {
    const int L = chunk_layer(cc), C = cc - chunk_first(L);
    return b16_layer_off_kib(L) + C * tpc(L) * b16_ks(L);
}

constexpr int LDS_WBUF = 40 * 1024;
constexpr int LDS_BIAS = 0;
constexpr int LDS_W0 = 10 * 1024;
constexpr int LDS_POSD = LDS_W0 + 2 * LDS_WBUF;
constexpr int LDS_POSX = LDS_POSD + WAVES * NCB * 1024;
constexpr int LDS_TOTAL = LDS_POSX + WAVES * NCB * 2048;
// COMP only: the sample ring behind everything else (16 B + 4 B per sample)
constexpr int RING_PTS = 1024;
constexpr int LDS_RING_RAW = LDS_TOTAL;
constexpr int LDS_RING_T = LDS_RING_RAW + RING_PTS * 16;
constexpr int LDS_TOTAL_COMP = LDS_RING_T + RING_PTS * 4;
constexpr int COMP_MAX_N = RING_PTS - TILE_PTS;      // an unfinished ray plus one more tile must fit
static_assert(COMP_MAX_N == FUSED_RENDER_MAX_N, "api.hip routes by this limit");
static_assert(B16_BIAS_FLOATS * 4 <= LDS_W0, "bias table");
static_assert(LDS_TOTAL_COMP <= 160 * 1024 && NUM_CHUNKS % 2 == 0, "LDS budget / parity");
static_assert((RING_PTS & (RING_PTS - 1)) == 0 && RING_PTS % TILE_PTS == 0, "ring indexing");

static_assert(ACT_TILE_PTS == TILE_PTS && MASK_TILE_PTS == TILE_PTS, "activation blocks and mask tiles are the kernel's tiles");

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lds_void;
template <class T>
__device__ __forceinline__ T lds_load(unsigned base, int imm) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T*>(
        reinterpret_cast<lds_char*>(0) + base + imm);
}
template <class T>
__device__ __forceinline__ void lds_store(unsigned base, int imm, const T& v) {
    *reinterpret_cast<__attribute__((address_space(3))) T*>(
        reinterpret_cast<lds_char*>(0) + base + imm) = v;
}

struct Ctx {
    __amdgpu_buffer_rsrc_t wrsrc;
    unsigned wave_goff, lane16;
    unsigned b_wread[2];            // weight buffer p + lane*16
    unsigned s_wdst[2];             // this wave's DMA piece in weight buffer p (wave-uniform)
    unsigned b_bias;                // (lane>>4)*16
    unsigned b_posx, b_posd;
    int wave, lane;
};

struct State {
    ex8 X[NCB][8], Y[NCB][8];        // [column block][k-step of 32]
    f32x4 pend[NCB][2];               // [column block][tile of the pending pair]
    float sigma[NCB], rgb[NCB][3];
    bool bad;                         // range guard: a non-finite accumulator was seen (flag_nonfinite)
    unsigned posd_off[NCB];           // LDS address of this lane's direction fragment (stage_inputs)
    // training forward only (SAVE): where this lane's activations go
    char* acts;
    long long P;
    long long tile;                   // tile index (uniform)
    int loff[NCB];                    // block_lane_offset(lane>>4, point in tile): this lane's granule of
                                      // fragment 0 in the tile's activation block; LOFF_INVALID past the end
    unsigned mb[NCB][2];              // ReLU mask bits being collected [column block][pair group]
    float amax[4];                    // 8-bit storage form: running maximum of the fragment group being finished, by (layer, group) parity
    long long mask_tile;              // byte offset of this tile's dword 0 of layer 0 (nerf_layout::mask_offset_bytes), uniform
    // fused render only (COMP)
    long long p_end;                  // one past this workgroup's last point (uniform)
    unsigned ring_q0;                 // ring slot of the tile's point 0 (uniform)
    struct WFrag* wf;                 // the coming chunk's first weight fragments (outlive a tile)
};
// The first AHEAD weight fragments and the first bias vector of the chunk that runs next, requested
// right behind the barrier that publishes its buffer -- which the inference kernels place three
// fragments BEFORE the end of the previous chunk, so the LDS round trip of these reads is covered by
// that chunk's last six MFMAs instead of idling the matrix pipe at every chunk start.
struct WFrag {
    ex8 a[4];
    f32x4 bias0;
};

template <bool RELU>
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f32x2 v = {a, b};
    const ex2 r = __builtin_convertvector(v, ex2);
    if constexpr (RELU) {
        const s16x2 z = {0, 0};
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, r), z));
    } else {
        return __builtin_bit_cast(unsigned, r);
    }
}

template <int CC>
struct Stage {
    static constexpr int NEXT = (CC + 1) % NUM_CHUNKS;
    static constexpr int PIECES = (chunk_kib(NEXT) + WAVES - 1) / WAVES;
    static constexpr int SRC_OFF = chunk_off_kib(NEXT) * 1024;
    static __device__ __forceinline__ void issue_piece(const Ctx& c, int p) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            c.wrsrc,
            reinterpret_cast<lds_void*>(reinterpret_cast<lds_char*>(0) + c.s_wdst[NEXT & 1] + p * (WAVES * 1024)),
            16, c.lane16, c.wave_goff + (SRC_OFF + p * WAVES * 1024), 0, 0);
    }
    static __device__ __forceinline__ void issue(const Ctx& c) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) issue_piece(c, p);
    }
};

// One of the 8 pieces of the epilogue of row-tile pair Q of layer L (tiles
// 2Q, 2Q+1; both column blocks): piece i -> column block i>>2, word i&3 of the
// next layer's fragment Q.  Heads: (L8, Q=8) is the lone sigma tile, L10 the
// rgb tile.
template <int L, int Q, int SAVE = 0>
__device__ __forceinline__ void epilogue_piece(int i, const f32x4 (&acc)[NCB][2], ex8 (&dst)[NCB][8], State& st) {
    constexpr LayerDesc D = layer_desc(L);
    const int cb = i >> 2, j2 = i & 3;   // i in [0, 4*NCB)
    if constexpr (Q == 0 && L >= 1 && L <= 9) {
        // Range guard.  If ANY input feature of this layer is inf (an fp16 activation beyond 65504) every one of its
        // rows sums w * inf: +-inf, or NaN with two of them -- so one accumulator element per point tells.  (The
        // outputs alone do not: the integer ReLU below turns a NaN with the sign bit set into 0, and a layer whose
        // rows are all NaN comes out as all zeros, finite from there on.)  One compare per column block and layer.
        if (j2 == 0) st.bad |= __builtin_amdgcn_classf(acc[cb][0][0], 0x207);      // sNaN | qNaN | -inf | +inf
    }
    if constexpr (L == 10) {
        if (j2 == 0) { st.rgb[cb][0] = acc[cb][0][0]; st.rgb[cb][1] = acc[cb][0][1]; st.rgb[cb][2] = acc[cb][0][2]; }
    } else if constexpr (L == 8 && Q == 8) {
        if (j2 == 0) st.sigma[cb] = acc[cb][0][0];
    } else {
        u32x4 w = __builtin_bit_cast(u32x4, dst[cb][Q]);
        w[j2] = pack2<D.relu != 0>(acc[cb][j2 >> 1][2 * (j2 & 1)], acc[cb][j2 >> 1][2 * (j2 & 1) + 1]);
        dst[cb][Q] = __builtin_bit_cast(ex8, w);
        if constexpr (SAVE == 2) {
            // 8-bit storage form: magnitudes are collected word by word over a group of four fragments (128 features); when
            // the group's last fragment is complete in both column blocks the wave converts and writes all four under one
            // exponent (nerf_device.h store_group_f8).  The accumulator is picked by (layer, group) parity: the pending pair
            // of the previous layer and this layer's first pair can be in flight together.
            constexpr int GS = (L & 1) * 2 + ((Q >> 2) & 1);
            st.amax[GS] = f8_absmax<D.relu == 0>(((Q & 3) == 0 && i == 0) ? 0.f : st.amax[GS], acc[cb][j2 >> 1][2 * (j2 & 1)],
                                                 acc[cb][j2 >> 1][2 * (j2 & 1) + 1]);
            if constexpr ((Q & 3) == 3) {
                if (i == 4 * NCB - 1) {
                    static_assert(NCB == 2, "store_group_f8 takes the two column blocks of a wave");
                    constexpr int Q0 = Q - 3;
                    char* tb = st.acts + (f8_offset_bytes(L, st.P) + st.tile * F8_BLOCK_BYTES);
                    char* sp = st.acts + (f8_scale_offset_bytes(L, st.P) + st.tile * 64);
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tb, 0, (int)F8_BLOCK_BYTES, 0x00020000);
                    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(sp, 0, 64, 0x00020000);
                    const u32x4 g0[4] = {__builtin_bit_cast(u32x4, dst[0][Q0]), __builtin_bit_cast(u32x4, dst[0][Q0 + 1]),
                                         __builtin_bit_cast(u32x4, dst[0][Q0 + 2]), __builtin_bit_cast(u32x4, dst[0][Q0 + 3])};
                    const u32x4 g1[4] = {__builtin_bit_cast(u32x4, dst[1][Q0]), __builtin_bit_cast(u32x4, dst[1][Q0 + 1]),
                                         __builtin_bit_cast(u32x4, dst[1][Q0 + 2]), w};
                    store_group_f8<2>(rs, st.loff[0], Q0 * 8192, rss, (int)(threadIdx.x & 63), (int)(threadIdx.x >> 6) * 8 + Q0,
                                      g0, g1, st.amax[GS]);
                }
            }
        }
        if constexpr (SAVE) {
            // the fragment is complete: write this lane's 2 x 4 features of layer L's output
            // (features 32Q+4g.. and 32Q+16+4g.. of its point) for the backward pass
            if (SAVE == 1 && j2 == 3) {
                // Buffer stores into this (layer, tile)'s point-blocked block (nerf_layout.h),
                // unconditional so the vector-memory instruction count per chunk is a constant
                // (chunk_barrier); lanes past the last point carry an offset outside num_records.
                // The lane holds two 8-byte pieces (features 32Q+4g.. and 32Q+16+4g..):
                // v_permlane16_swap trades one with the neighbouring 16-lane row (g ^ 1) -- even g
                // ends up with [its first piece | g+1's first piece], odd g with [g-1's second piece |
                // its second piece], i.e. one whole 16-byte granule (chunk 4Q + swapped_chunk(g)).
                // The 16 lanes of a quarter-wave then write 256 contiguous bytes.
                char* tb = st.acts + (act_offset_bytes(L, st.P) + st.tile * ACT_BLOCK_BYTES);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tb, 0, (int)ACT_BLOCK_BYTES, 0x00020000);
                store_granule<2>(rs, st.loff[cb], Q * 16384, w);
            }
            if constexpr (D.relu != 0) {
                // ReLU mask for the backward pass: one bit per feature (post-ReLU bf16 != 0), collected
                // per group of 4 pairs and written as one coalesced dword per thread
                static_assert(NCB == 2, "mask layout: two column blocks per wave");
                // t = {low != 0, high != 0} as 0/1 halves: one packed unsigned min with {1, 1} (hipcc
                // lowers the generic form to two compares, two selects and a permute)
                const unsigned wj = w[j2];
                unsigned t;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(wj), "s"(0x00010001u));
                const int pos = (Q & 3) * 4 + j2;
                st.mb[cb][Q >> 2] = pos == 0 ? t : ((t << pos) | st.mb[cb][Q >> 2]);
                if (pos == 15) {
                    // wave-uniform 64-bit base + this thread's 32-bit offset (scalar base, one offset VGPR)
                    char* mp = st.acts + (st.mask_tile + ((long long)L * mask_tiles(st.P) * 4 + (cb * 2 + (Q >> 2))) * 2048);
                    *reinterpret_cast<unsigned*>(mp + (unsigned)(threadIdx.x * 4)) = st.mb[cb][Q >> 2];
                }
            }
        }
    }
}

// vector-memory instructions epilogue_piece<L, Q, SAVE> issues over its 4*NCB pieces
template <int SAVE>
__host__ __device__ constexpr int pair_vmem_ops(int L, int Q) {
    if (!SAVE || L < 0 || L == 10 || (L == 8 && Q == 8)) return 0;
    const int masks = (layer_desc(L).relu != 0 && (Q & 3) == 3) ? NCB : 0;  // a mask dword per block behind every fourth pair
    // bf16 form: one activation store per block; 8-bit form: a group of four fragments at once (4 stores + 1 exponent dword)
    return (SAVE == 2 ? ((Q & 3) == 3 ? F8_GROUP + 1 : 0) : NCB) + masks;
}
template <int SAVE>
__host__ __device__ constexpr int pair_mask_ops(int L, int Q) {
    if (!SAVE || L < 0 || L == 10 || (L == 8 && Q == 8)) return 0;
    return (layer_desc(L).relu != 0 && (Q & 3) == 3) ? NCB : 0;
}

// the same summed over the pairs a chunk finishes itself (pairs P0 .. P0 + N - 1 of layer L)
template <int SAVE>
__host__ __device__ constexpr int chunk_pair_vmem_ops(int L, int P0, int N) {
    int n = 0;
    for (int j = 0; j < N; ++j) n += pair_vmem_ops<SAVE>(L, P0 + j);
    return n;
}
template <int SAVE>
__host__ __device__ constexpr int vmem_before_barrier(int L, int PL, int PQ, int pair0, int npair_in, int pend_m0, int pend_per,
                                                      int pair_m0, int mt, int m_limit) {
    int n = 0;
    // Column block cb's last piece (4 cb + 3) carries its mask dword and, in the bf16 form, its activation store; in the
    // 8-bit form the wave's column blocks are converted and written together -- a whole group of four fragments -- by
    // the last piece of the group's last pair.
    const int last = 4 * NCB - 1;
    if (PL >= 0) {
        const int masks = pair_mask_ops<SAVE>(PL, PQ), data = pair_vmem_ops<SAVE>(PL, PQ) - masks;
        for (int cb = 0; cb < NCB; ++cb) {
            const int own = 4 * cb + 3;
            if (pend_m0 + own / pend_per < m_limit) n += masks / NCB + (SAVE == 2 ? 0 : data / NCB);
        }
        if (SAVE == 2 && pend_m0 + last / pend_per < m_limit) n += data;
    }
    for (int j = 0; j < npair_in; ++j) {
        const int masks = pair_mask_ops<SAVE>(L, pair0 + j), data = pair_vmem_ops<SAVE>(L, pair0 + j) - masks;
        for (int cb = 0; cb < NCB; ++cb) {
            const int own = 4 * cb + 3;
            if (pair_m0 + j * 2 * mt + own < m_limit) n += masks / NCB + (SAVE == 2 ? 0 : data / NCB);
        }
        if (SAVE == 2 && pair_m0 + j * 2 * mt + last < m_limit) n += data;
    }
    return n;
}
// epilogue piece `i` of in-chunk pair j (a compile-time pair index is needed: dispatch over the few values)
template <int L, int P0, int N, int SAVE, int J = 0>
__device__ __forceinline__ void in_chunk_epilogue(int j, int i, const f32x4 (&acc)[NCB][2], ex8 (&dst)[NCB][8], State& st) {
    if constexpr (J < N) {
        if (j == J) epilogue_piece<L, P0 + J, SAVE>(i, acc, dst, st);
        else in_chunk_epilogue<L, P0, N, SAVE, J + 1>(j, i, acc, dst, st);
    }
}

// ---- one chunk: NT 16-row tiles of layer L starting at tile C * tpc(L) -------------------
// PL/PQ: layer / pair of the pending accumulators handed over by the previous chunk.
template <int L, int C, int PL, int PQ, int SAVE>
__device__ __forceinline__ void chunk_step(const Ctx& c, State& st, ex8 (&in)[NCB][8], ex8 (&out)[NCB][8]) {
    constexpr LayerDesc D = layer_desc(L);
    constexpr int KS_CHAIN = D.chain_k / 32;
    constexpr int KS_EXTRA = D.extra_slots / 32;
    constexpr int KS = KS_CHAIN + KS_EXTRA;
    constexpr int CC = chunk_first(L) + C;
    constexpr int NT = chunk_tiles(CC);
    constexpr int RT0 = C * tpc(L);
    constexpr int F = NT * KS;                      // weight fragments (each feeds 2 MFMAs)
    constexpr int AHEAD = 4;                         // weight fragments in flight ahead of their MFMAs (2..8 measure alike)
    constexpr int BIAS_OFF = LDS_BIAS + (b16_bias_off(L) + 16 * RT0) * 4;
    constexpr int XBLK = D.extra_kind == 1 ? 2048 : 1024;
    // chunk-linear MFMA index m = (t*KS + ks)*2 + cb
    constexpr int MT = NCB * KS;                      // MFMAs per row tile
    constexpr int PEND_M0 = (L == 10) ? 0 : (NT * MT >= 4 * NCB + 4 ? 2 : 0);
    constexpr int PEND_PER = (L == 10) ? 2 : 1;     // pieces per MFMA for the pending pair
    // Row-tile pairs of this chunk: pair j (tiles 2j, 2j+1; layer pair PAIR0 + j) gets its epilogue in the shadow
    // of the MFMAs of pair j + 1, starting at MFMA PAIR_M0 + j * 2 MT; the last pair is handed to the next chunk
    constexpr int PAIR0 = RT0 / 2;
    constexpr int NPAIR_IN = NT >= 4 ? NT / 2 - 1 : 0;
    constexpr int PAIR_M0 = 2 * MT + (MT >= 4 * NCB + 2 ? 2 : 0);
    static_assert(NT < 4 || NT % 2 == 0, "whole pairs per chunk");
    // a pending pair of the PREVIOUS layer is this layer's k-step PQ, first read by MFMA 2*PQ
    static_assert(PL < 0 || PL == L || (PL == 8 && PQ == 8) || NCB * PQ >= PEND_M0 + 4 * NCB / PEND_PER,
                  "pending pair finished too late");
    const unsigned wb = c.b_wread[CC & 1];
    const unsigned xb = c.b_posx;
    // the chunk that runs next (cyclic: the last chunk of a tile prefetches the first one of the next tile)
    constexpr int NCC = (CC + 1) % NUM_CHUNKS;
    constexpr int NL = chunk_layer(NCC);
    constexpr int NF = chunk_tiles(NCC) * (layer_desc(NL).chain_k / 32 + layer_desc(NL).extra_slots / 32);
    constexpr int NBIAS_OFF = LDS_BIAS + (b16_bias_off(NL) + 16 * (NCC - chunk_first(NL)) * tpc(NL)) * 4;
    const unsigned nwb = c.b_wread[NCC & 1];
    // Where the chunk's barrier sits, as a fragment index: TAIL fragments before the end of the chunk.
    constexpr int TAIL = 3;
    constexpr int FB = F <= TAIL ? F : F - TAIL;
    // The training forward's counted wait: vector-memory instructions this wave issues between its DMA pieces
    // (first in the chunk) and the barrier, i.e. the activation / mask stores of the epilogue pieces that sit in
    // front of MFMA FB * NCB (piece 4 cb + 3 of a pair carries column block cb's stores); the ones behind the
    // barrier are older than the next chunk's DMA and need no count.
    constexpr int VMEM_N = vmem_before_barrier<SAVE>(L, PL, PQ, PAIR0, NPAIR_IN, PEND_M0, PEND_PER, PAIR_M0, MT, FB * NCB);
    WFrag& wf = *st.wf;
    auto barrier_and_prefetch = [&]() {
        // Every fragment read of this chunk has been issued at least two fragment slots ago (AHEAD = 4,
        // TAIL = 3): lgkmcnt(0) is free, and it makes the buffer reusable -- no wave reads it after its
        // barrier.  vmcnt: this wave's LDS-DMA pieces of the next chunk (issued first in this chunk) have
        // landed; the training forward's stores behind them may fly on.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        chunk_barrier<VMEM_N>();
#pragma unroll
        for (int f = 0; f < AHEAD && f < NF; ++f) wf.a[f] = lds_load<ex8>(nwb, f * 1024);
        wf.bias0 = lds_load<f32x4>(c.b_bias, NBIAS_OFF);
        __builtin_amdgcn_sched_barrier(0);
    };

    // DMA of the next chunk: the training forward issues all pieces first (its counted vmcnt assumes every
    // store of the chunk behind them); the inference kernels, whose first fragments are already in
    // registers, start their MFMAs at once and issue one piece every SPREAD MFMAs -- provided the last
    // piece still goes out well before the MFMA in front of which the barrier publishes that buffer
    // (a piece issued behind the barrier would be read by the prefetch before it has landed)
    constexpr int SPREAD = 4;
    constexpr bool DMA_SPREAD = !SAVE && 1 + SPREAD * (Stage<CC>::PIECES - 1) + 8 <= FB * NCB;
    if constexpr (!DMA_SPREAD) Stage<CC>::issue(c);
    __builtin_amdgcn_sched_barrier(0);   // every other vector-memory instruction of the chunk stays behind the DMA

    ex8 a[AHEAD];
#pragma unroll
    for (int f = 0; f < AHEAD && f < F; ++f) a[f] = wf.a[f];       // requested behind the previous barrier
    ex8 bx[NCB][KS_EXTRA > 0 ? KS_EXTRA : 1];
    if constexpr (KS_EXTRA > 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int e = 0; e < KS_EXTRA; ++e)
                bx[cb][e] = D.extra_kind == 1 ? lds_load<ex8>(xb, cb * XBLK + e * 1024) : lds_load<ex8>(st.posd_off[cb], e * 1024);
    }
    f32x4 acc[NCB][NT];
    // register i of lane group g is row 16*rt + 4g + i: one 16-B bias read per tile
    acc[0][0] = wf.bias0;
    for (int cb = 1; cb < NCB; ++cb) acc[cb][0] = acc[0][0];
    __builtin_amdgcn_sched_barrier(0);

    // Register lifetimes against the MFMA write-after-read hazards.  The allocator hands the registers an
    // MFMA has just read for the last time (its weight fragment, and the old accumulator: D != C in the
    // VGPR form) to the very next definition -- the following ds_read or cvt_pk -- and the hazard
    // recognizer then puts 2-4 wait states between the two: 617 s_nop per tile, ~1700 cycles per wave,
    // in a stream whose issue time is what bounds the kernel.  Empty asm uses keep a fragment alive for one
    // more fragment slot and an accumulator for one more MFMA, so the registers that come free were last
    // read two instructions ago: 8 more live VGPRs, 190 s_nop per tile, -1.7 ... 2.1 % time.
    ex8 as_prev = a[0];
    f32x4 c_prev = acc[0][0];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int f = t * KS + ks;
            if (f == FB) barrier_and_prefetch();
            const ex8 as = a[f % AHEAD];
            if (f + AHEAD < F) a[f % AHEAD] = lds_load<ex8>(wb, (f + AHEAD) * 1024);
            if (t + 1 < NT && ks == KS / 2) {
                acc[0][t + 1 < NT ? t + 1 : 0] = lds_load<f32x4>(c.b_bias, BIAS_OFF + 64 * (t + 1));
                for (int cb = 1; cb < NCB; ++cb) acc[cb][t + 1 < NT ? t + 1 : 0] = acc[0][t + 1 < NT ? t + 1 : 0];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int m = f * NCB + cb;
                ex8 bs;
                if (ks < KS_CHAIN) bs = in[cb][ks < KS_CHAIN ? ks : 0];
                else bs = bx[cb][KS_EXTRA > 0 ? (ks - KS_CHAIN < KS_EXTRA ? ks - KS_CHAIN : 0) : 0];
                const f32x4 c_old = acc[cb][t];
                acc[cb][t] = NERF_MFMA(as, bs, c_old, 0, 0, 0);
                asm volatile("" :: "v"(c_prev));
                c_prev = c_old;
                if constexpr (DMA_SPREAD) {
                    if (m % SPREAD == 1 && m / SPREAD < Stage<CC>::PIECES) Stage<CC>::issue_piece(c, m / SPREAD);
                }
                // ---- epilogue pieces in this MFMA's shadow
                if constexpr (PL >= 0) {
                    if (m >= PEND_M0 && m < PEND_M0 + 4 * NCB / PEND_PER) {
#pragma unroll
                        for (int k = 0; k < PEND_PER; ++k) {
                            const int i = (m - PEND_M0) * PEND_PER + k;
                            if constexpr (PL == L) epilogue_piece<PL, PQ, SAVE>(i, st.pend, out, st);
                            else epilogue_piece<PL, PQ, SAVE>(i, st.pend, in, st);
                        }
                    }
                }
                if constexpr (NPAIR_IN > 0) {
                    const int j = (m - PAIR_M0) / (2 * MT), pm = (m - PAIR_M0) - j * (2 * MT);
                    if (m >= PAIR_M0 && j < NPAIR_IN && pm < 4 * NCB) {
                        f32x4 pr[NCB][2];
                        for (int q_ = 0; q_ < NCB; ++q_) { pr[q_][0] = acc[q_][2 * j]; pr[q_][1] = acc[q_][2 * j + 1]; }
                        in_chunk_epilogue<L, PAIR0, NPAIR_IN, SAVE>(j, pm, pr, out, st);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" :: "v"(as_prev));
            as_prev = as;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("" :: "v"(as_prev));
    asm volatile("" :: "v"(c_prev));
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        st.pend[cb][0] = acc[cb][NT >= 2 ? NT - 2 : 0];
        st.pend[cb][1] = acc[cb][NT - 1];
    }
    if constexpr (FB == F) barrier_and_prefetch();
}

__host__ __device__ constexpr int prev_layer(int L, int C) { return C > 0 ? L : L - 1; }
__host__ __device__ constexpr int prev_pair(int L, int C) {
    // pending pair when chunk (L, C) starts: same layer -> the last pair of chunk C-1; else the previous
    // layer's last pair (L8 ends with its lone sigma tile, marked as pair 8)
    return C > 0 ? C * tpc(L) / 2 - 1 : (L > 0 ? (L - 1 == 8 ? 8 : b16_mt(L - 1) / 2 - 1) : 0);
}

template <int L, int SAVE, int... Cs>
__device__ __forceinline__ void run_layer_seq(const Ctx& c, State& st, ex8 (&in)[NCB][8], ex8 (&out)[NCB][8],
                                              std::integer_sequence<int, Cs...>) {
    (chunk_step<L, Cs, prev_layer(L, Cs), prev_pair(L, Cs), SAVE>(c, st, in, out), ...);
}
template <int L, int SAVE>
__device__ __forceinline__ void run_layer(const Ctx& c, State& st, ex8 (&in)[NCB][8], ex8 (&out)[NCB][8]) {
    run_layer_seq<L, SAVE>(c, st, in, out, std::make_integer_sequence<int, layer_chunks(L)>{});
}

// sin(2 pi (2^level q + trig/4)) with a per-lane level / trig
__device__ __forceinline__ float enc_lane(TwoF q, int idx) {
    const float sc = __builtin_amdgcn_ldexpf(1.0f, idx >> 1);
    const float fr = __builtin_amdgcn_fractf(q.hi * sc) + q.lo * sc + ((idx & 1) ? 0.25f : 0.f);
    return __builtin_amdgcn_sinf(fr);
}

// SAVE (the training forward, launched in rays mode) also serves Nerf.forward(v) with gradients:
// a.pts != NULL switches the point fetch at run time, so training needs no third instantiation.
template <bool RAYS, int SAVE, bool COMP>
__device__ __forceinline__ void stage_inputs(const Ctx& c, const MlpArgs& a, long long tile_base, State& st) {
    const int col = c.lane & 15, g = c.lane >> 4;
    // Counter-RNG jitter: the four lane groups of a point would each evaluate the same Philox
    // (40 quarter-rate integer multiplies), once per column block.  Instead every lane evaluates
    // it once, for point (lane & 31) of the wave's 32, and the lanes fetch their two points'
    // draws with a wave shuffle: the same numbers for half the work.
    float u_mine = 0.f;
    bool dev_rng = false;
    long long b0 = 0;                                   // divmod(tile_base, N): wave-uniform, once per tile
    int r0 = 0;
    if constexpr (RAYS) {
        b0 = tile_base / a.N;
        r0 = (int)(tile_base - b0 * a.N);
    }
    const long long p_end = COMP ? st.p_end : a.P;
    const int last_local = (int)(p_end - 1 - tile_base);  // lanes past the end use the last point (results dropped)
    // The direction features (posd: 24 sines / cosines, 3 raw coordinates) belong to the RAY:
    // in rays mode they are evaluated once per ray of the tile -- one LDS slot of 64 B per ray, value (group g, slot s)
    // by thread 32 * ray + 8 g + s, two rays per tile at N = 128 -- instead of once per sample, and with them goes
    // the per-sample normalisation of the direction (a square root and three exact divisions).  Every lane then
    // reads its ray's 16-byte fragment (a broadcast within the 16 points of a column block).  Same operations per
    // value as the per-sample form: bit-identical features.  (The training forward keeps the per-sample form: it
    // also serves explicit points, whose directions are per point.)
    constexpr bool RAY_POSD = RAYS && !SAVE;
    if constexpr (RAY_POSD) {
        const int top = last_local < TILE_PTS - 1 ? last_local : TILE_PTS - 1;
        const int nrays = (int)(split_point(b0, r0, top, a.N).b - b0) + 1;          // rays this tile touches (uniform)
        for (int idx = threadIdx.x; idx < nrays * 32; idx += WAVES * 64) {
            const int ray = idx >> 5, slot = idx & 31, gg = slot >> 3, sl = slot & 7;
            const float* rp = a.rays + (b0 + ray) * 6 + 3;
            const float dx = rp[0], dy = rp[1], dz = rp[2];
            const float nrm = norm3(dx, dy, dz);              // as fetch_point_rays normalises: torch.norm bit for bit
            float val = 0.f;
            if (sl < 6) {                                     // posd_col_f32: per coordinate the pair (level g, trig)
                const int cd = sl >> 1;
                const float dc = __fdiv_rn(cd == 0 ? dx : cd == 1 ? dy : dz, nrm);
                val = enc_lane(to_revolutions(dc), 2 * gg + (sl & 1));
            } else if (sl == 6 && gg < 3) {
                val = __fdiv_rn(gg == 0 ? dx : gg == 1 ? dy : dz, nrm);
            }
            lds_store<elem_t>(LDS_POSD + ray * 64 + slot * 2, 0, (elem_t)val);
        }
    }
    if constexpr (RAYS && NCB == 2) {
        dev_rng = (a.flags & NERF_FLAG_DEVICE_RNG) && !(a.flags & NERF_FLAG_TS_GIVEN);
        if (dev_rng) {
            int lm = c.wave * 32 + (c.lane & 31);
            if (lm > last_local) lm = last_local;
            u_mine = device_rng_uniform(a, split_point(b0, r0, lm, a.N));
        }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        long long p = tile_base + c.wave * (16 * NCB) + cb * 16 + col;
        const bool valid = p < p_end;
        st.loff[cb] = valid ? block_lane_offset(g, c.wave * (16 * NCB) + cb * 16 + col) : LOFF_INVALID;
        if constexpr (SAVE == 2) {
            // 8-bit form: after store_fragment_f8's transpose the lane writes the granule of point 16 (lane >> 5) + col of
            // the wave, not of its own two points
            if (cb == 0)
                st.loff[0] = tile_base + c.wave * 32 + 16 * (c.lane >> 5) + col < p_end ? f8_lane_offset(c.lane, c.wave) : LOFF_INVALID;
        }
        if (!valid) p = p_end - 1;
        PointIn pt;
        st.posd_off[cb] = c.b_posd + cb * 1024;              // per-sample form: this lane's own fragment
        if constexpr (RAYS) {
            const float u_cb = (NCB == 2) ? __shfl(u_mine, cb * 16 + col) : 0.f;
            int lp = c.wave * (16 * NCB) + cb * 16 + col;
            if (lp > last_local) lp = last_local;
            if (SAVE && a.pts) {
                pt = fetch_point_pts(a, p);
            } else {
                const RaySample rs = split_point(b0, r0, lp, a.N);
                if constexpr (RAY_POSD) st.posd_off[cb] = LDS_POSD + (unsigned)(rs.b - b0) * 64 + g * 16;
                pt = fetch_point_rays<!RAY_POSD>(a, p, rs, u_cb, dev_rng);
                if constexpr (COMP) {
                    if (valid && g == 0)
                        lds_store<float>(((st.ring_q0 + c.wave * (16 * NCB) + cb * 16 + col) & (RING_PTS - 1)) * 4, LDS_RING_T, pt.t);
                }
                if (valid && g == 0 && a.ts_out) a.ts_out[p] = pt.t;
            }
        } else {
            pt = fetch_point_pts(a, p);
        }
        {   // posx: 16 slots per lane group (nerf_layout::posx_col_f32)
            const float xyz[3] = {pt.x, pt.y, pt.z};
            float v[16];
#pragma unroll
            for (int cd = 0; cd < 3; ++cd) {
                const TwoF q = to_revolutions(xyz[cd]);
#pragma unroll
                for (int jj = 0; jj < 5; ++jj) v[cd * 5 + jj] = enc_lane(q, 5 * g + jj);
            }
            v[15] = g == 0 ? pt.x : g == 1 ? pt.y : g == 2 ? pt.z : 0.f;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                u32x4 r;
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = pack2<false>(v[8 * e + 2 * i], v[8 * e + 2 * i + 1]);
                lds_store<u32x4>(c.b_posx, cb * 2048 + e * 1024, r);
            }
        }
        if constexpr (!RAY_POSD) {   // posd: 8 slots per lane group (nerf_layout::posd_col_f32)
            const float dd[3] = {pt.d1, pt.d2, pt.d3};
            float v[8];
#pragma unroll
            for (int cd = 0; cd < 3; ++cd) {
                const TwoF q = to_revolutions(dd[cd]);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) v[cd * 2 + jj] = enc_lane(q, 2 * g + jj);
            }
            v[6] = g == 0 ? pt.d1 : g == 1 ? pt.d2 : g == 2 ? pt.d3 : 0.f;
            v[7] = 0.f;
            u32x4 r;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = pack2<false>(v[2 * i], v[2 * i + 1]);
            lds_store<u32x4>(c.b_posd, cb * 1024, r);
        }
    }
}

// one tile: prologue (sampling / RNG / encoding into LDS) + the 11 layers; leaves st.rgb / st.sigma
template <bool RAYS, int SAVE, bool COMP>
__device__ __forceinline__ void run_tile(const Ctx& c, const MlpArgs& a, long long tile_base, State& st) {
    stage_inputs<RAYS, SAVE, COMP>(c, a, tile_base, st);
    run_layer<0, SAVE>(c, st, st.X, st.X);
    run_layer<1, SAVE>(c, st, st.X, st.Y);
    run_layer<2, SAVE>(c, st, st.Y, st.X);
    run_layer<3, SAVE>(c, st, st.X, st.Y);
    run_layer<4, SAVE>(c, st, st.Y, st.X);
    run_layer<5, SAVE>(c, st, st.X, st.Y);
    run_layer<6, SAVE>(c, st, st.Y, st.X);
    run_layer<7, SAVE>(c, st, st.X, st.Y);
    run_layer<8, SAVE>(c, st, st.Y, st.X);
    run_layer<9, SAVE>(c, st, st.X, st.Y);
    run_layer<10, SAVE>(c, st, st.Y, st.X);
    epilogue_piece<10, 0>(0, st.pend, st.X, st);     // the rgb tile is still pending
    for (int cb_ = 1; cb_ < NCB; ++cb_) epilogue_piece<10, 0>(4 * cb_, st.pend, st.X, st);
}

// Sticky range flag (nerf_layout.h B16_STATUS_OFF): a point with a non-finite accumulator in layers 1..9 (epilogue_piece)
// or a non-finite (rgb, sigma) sets status word 0 behind the packed image.  The host wrapper reads it
// (utils/nets.py): the reference is fp32 and has no range limit (utils/nets.py:16-32), so an overflowing fp16 render
// must not pass silently.  A plain store of the constant 1 through the weight image's own buffer descriptor (every
// writer writes the same value: no atomic, no extra pointer kept live across the tile loop).
__device__ __forceinline__ void flag_nonfinite(const Ctx& c, bool bad) {
    if (bad) __builtin_amdgcn_raw_buffer_store_b32(1u, c.wrsrc, (int)(B16_STATUS_OFF + 4 * NERF_STATUS_WORD_NONFINITE), 0, 0);
}
__device__ __forceinline__ bool finite4(float x, float y, float z, float w) {
    // |v| < inf is false for inf and NaN; the sum is non-finite iff any term is (no finite sum of four floats overflows
    // unless a term is already beyond half of FLT_MAX -- which fp16 / bf16 MLP outputs of a usable network never are)
    return __builtin_fabsf(x) + __builtin_fabsf(y) + __builtin_fabsf(z) + __builtin_fabsf(w) < __builtin_inff();
}

struct RingSamples {                         // a ray's samples in the workgroup's LDS ring
    unsigned q0;                              // ring slot of its sample 0
    __device__ __forceinline__ float t(int i) const {
        return lds_load<float>(((q0 + (unsigned)i) & (RING_PTS - 1)) * 4, LDS_RING_T);
    }
    __device__ __forceinline__ f32x4 c(int i) const {
        return lds_load<f32x4>(((q0 + (unsigned)i) & (RING_PTS - 1)) * 16, LDS_RING_RAW);
    }
};

template <bool RAYS, int SAVE, bool COMP>
__device__ __forceinline__ void kernel_body(const MlpArgs& a, long long ntiles) {
    static_assert(!COMP || (RAYS && !SAVE), "the fused render is the rays-mode inference kernel");
    Ctx c;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    const char* img = reinterpret_cast<const char*>(a.packed);
    c.wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img), 0, (int)B16_IMAGE_BYTES, 0x00020000);
    c.wave_goff = c.wave * 1024;
    c.lane16 = c.lane * 16;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        c.b_wread[p] = LDS_W0 + p * LDS_WBUF + c.lane * 16;
        c.s_wdst[p] = LDS_W0 + p * LDS_WBUF + c.wave * 1024;
    }
    c.b_bias = (c.lane >> 4) * 16;
    c.b_posx = LDS_POSX + c.wave * (NCB * 2048) + c.lane * 16;
    c.b_posd = LDS_POSD + c.wave * (NCB * 1024) + c.lane * 16;

    {
        const float* bsrc = reinterpret_cast<const float*>(img + (long long)B16_WEIGHT_KIB * 1024);
        for (int i = threadIdx.x; i < B16_BIAS_FLOATS; i += WAVES * 64)
            lds_store<float>(i * 4, LDS_BIAS, bsrc[i]);
        Stage<NUM_CHUNKS - 1>::issue(c);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
    __syncthreads();
    WFrag wf;                                           // chunk 0's first fragments (chunk_step hands them on)
    {
        constexpr int F0 = chunk_tiles(0) * (layer_desc(0).chain_k / 32 + layer_desc(0).extra_slots / 32);
#pragma unroll
        for (int f = 0; f < 4 && f < F0; ++f) wf.a[f] = lds_load<ex8>(c.b_wread[0], f * 1024);
        wf.bias0 = lds_load<f32x4>(c.b_bias, LDS_BIAS + b16_bias_off(0) * 4);
    }

    if constexpr (!COMP) {
        for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const long long tile_base = tile * TILE_PTS;
            asm volatile("" : "+s"(c.wave_goff));
            State st;
            st.wf = &wf;
            st.acts = reinterpret_cast<char*>(a.acts);
            st.P = a.P;
            st.mask_tile = SAVE == 2 ? f8_mask_offset_bytes(0, tile, 0, a.P) : mask_offset_bytes(0, tile, 0, a.P);
            st.tile = tile;
            st.bad = false;
            run_tile<RAYS, SAVE, false>(c, a, tile_base, st);
            bool bad = st.bad;
            if (c.lane < 16) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const long long p = tile_base + c.wave * (16 * NCB) + cb * 16 + c.lane;
                    if (p < a.P) {
                        const f32x4 o = {st.rgb[cb][0], st.rgb[cb][1], st.rgb[cb][2], st.sigma[cb]};
                        *reinterpret_cast<f32x4*>(a.raw + p * 4) = o;
                        bad |= !finite4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
            flag_nonfinite(c, bad);
        }
    } else {
        // ---- fused render: this workgroup's contiguous range of rays, tile after tile ----------
        const long long B = a.P / a.N;
        const long long r_lo = (long long)blockIdx.x * B / gridDim.x, r_hi = ((long long)blockIdx.x + 1) * B / gridDim.x;
        const long long range_base = r_lo * a.N;
        const int n_pts = (int)((r_hi - r_lo) * a.N);                 // < 2^31: the launcher splits larger calls
        const nerf_composite::RayOut out{a.rgb, a.disp, a.alpha, a.acc, a.w, a.pixels};
        int next_ray = 0, n_complete = 0;                             // rays composited / completely in the ring (uniform)
        for (int q_tile = 0; q_tile < n_pts; q_tile += TILE_PTS) {
            const long long tile_base = range_base + q_tile;
            asm volatile("" : "+s"(c.wave_goff));
            State st;
            st.wf = &wf;
            st.acts = nullptr;
            st.P = a.P;
            st.mask_tile = 0;
            st.tile = 0;
            st.p_end = range_base + n_pts;
            st.ring_q0 = (unsigned)q_tile & (RING_PTS - 1);
            st.bad = false;
            run_tile<true, false, true>(c, a, tile_base, st);
            bool bad = st.bad;
            if (c.lane < 16) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const int local = c.wave * (16 * NCB) + cb * 16 + c.lane;
                    if (q_tile + local < n_pts) {
                        const f32x4 o = {st.rgb[cb][0], st.rgb[cb][1], st.rgb[cb][2], st.sigma[cb]};
                        lds_store<f32x4>(((st.ring_q0 + local) & (RING_PTS - 1)) * 16, LDS_RING_RAW, o);
                        bad |= !finite4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
            flag_nonfinite(c, bad);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                             // every wave's samples of this tile are in the ring
            asm volatile("" ::: "memory");
            const int done_q = q_tile + TILE_PTS < n_pts ? q_tile + TILE_PTS : n_pts;
            while ((n_complete + 1) * a.N <= done_q) ++n_complete;
            // composite when every wave has a ray, when the ring could not take another tile, or at the end
            if (n_complete - next_ray >= WAVES || done_q + TILE_PTS - next_ray * a.N > RING_PTS || done_q == n_pts) {
                for (int ray = next_ray + c.wave; ray < n_complete; ray += WAVES) {
                    const long long gray = r_lo + ray;
                    const float* d = a.rays + gray * 6 + 3;
                    const float dnorm = nerf_composite::unit_dir_norm(d[0], d[1], d[2], true);
                    const RingSamples src{(unsigned)(ray * a.N) & (RING_PTS - 1)};
                    nerf_composite::composite_ray(src, a.N, c.lane, dnorm, gray, out);
                }
                next_ray = n_complete;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                         // the next tile's prologue rewrites ring slots read above
                asm volatile("" ::: "memory");
            }
        }
    }
}

template <bool RAYS, bool SAVE, bool COMP>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void NERF_KERNEL(MlpArgs a, long long ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    (void)smem;
    kernel_body<RAYS, SAVE ? 1 : 0, COMP>(a, ntiles);
}
#ifndef NERF_HALF
// the training forward with the 8-bit storage form of the saved activations (nerf_layout.h; MlpArgs::flags bit
// NERF_FLAG_STORE_E4M3): a kernel of its own name, so the bf16-storage instantiations keep theirs
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void nerf_mlp_train_e4m3_kernel(MlpArgs a, long long ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    (void)smem;
    kernel_body<true, 2, false>(a, ntiles);
}
#endif

}  // namespace

extern "C" int NERF_LAUNCH(const MlpArgs* args, int rays_mode, hipStream_t stream) {
    (void)hipGetLastError();
    MlpArgs a = *args;
    if (a.P <= 0) return 0;
    const long long ntiles = (a.P + TILE_PTS - 1) / TILE_PTS;
    const int cus = device_cus();
    const long long grid = ntiles < cus ? ntiles : cus;
    hipError_t e = hipSuccess;
    const bool comp = a.rgb || a.disp || a.acc || a.alpha || a.w || a.pixels;
    if (comp) {
        // fused render: rays mode, inference; a ray plus one tile must fit the LDS ring, and a
        // workgroup's share of the points must fit an int
        if (!rays_mode || a.acts || a.N > COMP_MAX_N || a.P / grid + a.N >= (1ll << 31)) return -2;
        auto kern = NERF_KERNEL<true, false, true>;
        e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_TOTAL_COMP);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * 64), LDS_TOTAL_COMP, stream, a, ntiles);
        return (int)hipGetLastError();
    }
#ifdef NERF_HALF
    if (a.acts) return -2;                       // the training forward exists in bf16 only
    auto kern = rays_mode ? NERF_KERNEL<true, false, false> : NERF_KERNEL<false, false, false>;
#else
    if (a.acts && !rays_mode) return -2;          // the training forward is the rays-mode instantiation (a.pts selects points)
    if ((a.flags & NERF_FLAG_STORE_E4M3) && !a.acts) return -2;
    auto kern = a.acts ? ((a.flags & NERF_FLAG_STORE_E4M3) ? nerf_mlp_train_e4m3_kernel : NERF_KERNEL<true, true, false>)
                       : (rays_mode ? NERF_KERNEL<true, false, false> : NERF_KERNEL<false, false, false>);
#endif
    e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_TOTAL);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * 64), LDS_TOTAL, stream, a, ntiles);
    return (int)hipGetLastError();
}
