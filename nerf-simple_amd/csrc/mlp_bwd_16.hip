// mlp_bwd_16.hip -- training backward of the 12 dense layers, dX chain.
//
// Autograd through Nerf.forward (reference utils/nets.py:34-43) inside the
// training step (reference train.py:51-54), restricted to what the step needs:
// gradients w.r.t. the PARAMETERS.  This kernel computes, for a tile of 256
// points, every layer's pre-activation gradient dY_l = relu'(h_l) (.) (W_{l+1}^T
// dY_{l+1}) with the same on-chip chaining as the forward kernel
// (mlp_bf16_16.hip; read its header first): the accumulators of one backward
// layer, masked and converted to bf16, are the B operand of the next one, so the
// gradient w.r.t. activations never leaves the CU.  It reads d_raw[P,4] (from
// the compositor's backward) and the ReLU masks (one bit per feature, written by
// the training forward in this kernel's own register layout: nerf_layout.h) and
// writes dY_l (bf16, point-blocked like the saved activations).
// The weight gradients dW_l = dY_l^T X_l are plain GEMMs over the point
// dimension and run in dw_gemm.hip.
//
// 10 backward layers (nerf_layout::bwd_desc), 38 weight chunks per tile streamed
// L2 -> LDS by LDS-DMA exactly as in the forward.  HBM-bound by construction:
// 0.3 KB of mask bits read and ~5 KB of dY written per point.
#include "nerf_device.h"
#include <utility>

using namespace nerf_layout;

typedef __bf16 ex8 __attribute__((ext_vector_type(8)));
typedef __bf16 ex2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct BwdArgs {
    const float* d_raw;       // [P,4] = d loss / d [r,g,b,sigma]
    const void* image;        // backward weight image (nerf_amd_pack_weights precision 3)
    const char* acts;         // saved forward activations
    char* dys;                // out: pre-activation gradients, same layout as acts
    long long P;
    int store_e4m3;           // acts (its masks) and dys are in the 8-bit storage form (nerf_layout.h)
};

namespace {

constexpr int WAVES = 8;
constexpr int TILE_PTS = WAVES * 32;
constexpr int TPC = 4;

__host__ __device__ constexpr int layer_chunks(int b) { return (bwd_mt(b) + TPC - 1) / TPC; }
__host__ __device__ constexpr int chunk_first(int b) {
    int c = 0;
    for (int i = 0; i < b; ++i) c += layer_chunks(i);
    return c;
}
constexpr int NUM_CHUNKS = chunk_first(NUM_BWD);              // 38
__host__ __device__ constexpr int chunk_layer(int cc) {
    int b = 0;
    while (cc >= layer_chunks(b)) { cc -= layer_chunks(b); ++b; }
    return b;
}
__host__ __device__ constexpr int chunk_kib(int cc) { return TPC * bwd_ks(chunk_layer(cc)); }
__host__ __device__ constexpr int chunk_off_kib(int cc) {
    const int b = chunk_layer(cc), C = cc - chunk_first(b);
    return bwd_layer_off_kib(b) + C * TPC * bwd_ks(b);
}

constexpr int LDS_WBUF = 40 * 1024;
constexpr int LDS_W0 = 0;
constexpr int LDS_TOTAL = 2 * LDS_WBUF;
static_assert(NUM_CHUNKS % 2 == 0, "buffer parity must repeat per tile");

static_assert(ACT_TILE_PTS == TILE_PTS && MASK_TILE_PTS == TILE_PTS, "activation blocks and mask tiles are the kernel's tiles");

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lds_void;
template <class T>
__device__ __forceinline__ T lds_load(unsigned base, int imm) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T*>(
        reinterpret_cast<lds_char*>(0) + base + imm);
}

struct Ctx {
    __amdgpu_buffer_rsrc_t wrsrc;
    unsigned wave_goff, lane16;
    unsigned b_wread[2], s_wdst[2];
    int wave, lane;
};

struct State {
    ex8 X[2][8], Y[2][8];           // dY fragments, ping-pong
    f32x4 pend[2][2];               // pending pair accumulators [cb][tile]
    u32x4 mk[2];                    // ReLU mask dwords of backward layer B in mk[B & 1] (nerf_layout.h)
    long long mask_tile;            // byte offset of this tile's dword 0 of activation 0 (uniform)
    ex8 bx_rgb[2], bx_sig[2];       // custom k-steps built from d_raw
    const char* acts;
    char* dys;
    long long P, tile;
    int loff[2];                    // block_lane_offset(lane>>4, point in tile), LOFF_INVALID past the end
    float amax[4];                  // 8-bit storage form: running maximum of the fragment group being finished, by (layer, group) parity
};

__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ex2));
}

template <int CC>
struct Stage {
    static constexpr int NEXT = (CC + 1) % NUM_CHUNKS;
    static constexpr int PIECES = (chunk_kib(NEXT) + WAVES - 1) / WAVES;
    static constexpr int SRC_OFF = chunk_off_kib(NEXT) * 1024;
    static __device__ __forceinline__ void issue(const Ctx& c) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                c.wrsrc,
                reinterpret_cast<lds_void*>(reinterpret_cast<lds_char*>(0) + c.s_wdst[NEXT & 1] + p * (WAVES * 1024)),
                16, c.lane16, c.wave_goff + (SRC_OFF + p * WAVES * 1024), 0, 0);
    }
};

// the four mask dwords of forward activation A for this thread's 2 x 256 (point, feature) cells
template <int A>
__device__ __forceinline__ void load_mask(const State& st, u32x4& m) {
    const char* mp = st.acts + (st.mask_tile + (long long)A * mask_tiles(st.P) * 8192);      // wave-uniform
#pragma unroll
    for (int d = 0; d < 4; ++d) m[d] = *reinterpret_cast<const unsigned*>(mp + d * 2048 + (unsigned)(threadIdx.x * 4));
}

// piece i of the epilogue of pair Q of backward layer B: mask, convert, and (on the
// fragment's last word) store dY of forward layer 9-B
template <int B, int Q, bool F8>
__device__ __forceinline__ void epilogue_piece(int i, const f32x4 (&acc)[2][2], ex8 (&dst)[2][8], State& st) {
    constexpr BwdDesc D = bwd_desc(B);
    constexpr int LOUT = 9 - B;
    const int cb = i >> 2, j2 = i & 3;
    float v0 = acc[cb][j2 >> 1][2 * (j2 & 1)], v1 = acc[cb][j2 >> 1][2 * (j2 & 1) + 1];
    if constexpr (D.mask_act >= 0) {
        // bit -> all-ones / zero by a sign-extending 1-bit field extract, then AND
        const unsigned mw = st.mk[B & 1][cb * 2 + (Q >> 2)];
        const int pos = (Q & 3) * 4 + j2;
        v0 = __builtin_bit_cast(float, __builtin_bit_cast(int, v0) & ((int)(mw << (31 - pos)) >> 31));
        v1 = __builtin_bit_cast(float, __builtin_bit_cast(int, v1) & ((int)(mw << (15 - pos)) >> 31));
    }
    u32x4 w = __builtin_bit_cast(u32x4, dst[cb][Q]);
    w[j2] = pack2(v0, v1);
    dst[cb][Q] = __builtin_bit_cast(ex8, w);
    if constexpr (F8) {
        // 8-bit storage form: a group of four fragments (both column blocks) is converted and written under one exponent
        // when its last fragment is complete (nerf_device.h store_group_f8); accumulator by (layer, group) parity
        constexpr int GS = (B & 1) * 2 + ((Q >> 2) & 1);
        st.amax[GS] = f8_absmax<true>(((Q & 3) == 0 && i == 0) ? 0.f : st.amax[GS], v0, v1);
        if constexpr ((Q & 3) == 3) {
            if (i == 7) {
                constexpr int Q0 = Q - 3;
                char* tb = st.dys + (f8_offset_bytes(LOUT, st.P) + st.tile * F8_BLOCK_BYTES);
                char* sp = st.dys + (f8_scale_offset_bytes(LOUT, st.P) + st.tile * 64);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tb, 0, (int)F8_BLOCK_BYTES, 0x00020000);
                const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(sp, 0, 64, 0x00020000);
                const u32x4 g0[4] = {__builtin_bit_cast(u32x4, dst[0][Q0]), __builtin_bit_cast(u32x4, dst[0][Q0 + 1]),
                                     __builtin_bit_cast(u32x4, dst[0][Q0 + 2]), __builtin_bit_cast(u32x4, dst[0][Q0 + 3])};
                const u32x4 g1[4] = {__builtin_bit_cast(u32x4, dst[1][Q0]), __builtin_bit_cast(u32x4, dst[1][Q0 + 1]),
                                     __builtin_bit_cast(u32x4, dst[1][Q0 + 2]), w};
                store_group_f8<2>(rs, st.loff[0], Q0 * 8192, rss, (int)(threadIdx.x & 63), (int)(threadIdx.x >> 6) * 8 + Q0, g0, g1,
                                  st.amax[GS]);
            }
        }
    } else if (j2 == 3) {
        // one 16-byte granule per lane into the (layer, tile) block of dY, after trading 8-byte
        // pieces with lane group g ^ 1; unconditional buffer store (forward epilogue_piece)
        char* tb = st.dys + (act_offset_bytes(LOUT, st.P) + st.tile * ACT_BLOCK_BYTES);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tb, 0, (int)ACT_BLOCK_BYTES, 0x00020000);
        store_granule<2>(rs, st.loff[cb], Q * 16384, w);      // non-temporal: see store_granule
    }
}

template <int B, int C, int PB, int PQ, bool F8>
__device__ __forceinline__ void chunk_step(const Ctx& c, State& st, ex8 (&in)[2][8], ex8 (&out)[2][8]) {
    constexpr BwdDesc D = bwd_desc(B);
    constexpr int KS_CHAIN = D.chain_k / 32;
    constexpr int KS = KS_CHAIN + D.extra;
    constexpr int CC = chunk_first(B) + C;
    constexpr int NT = TPC;                         // every backward layer has a multiple of 4 row tiles
    constexpr int F = NT * KS;
    constexpr int AHEAD = 4;
    constexpr int MT = 2 * KS;                      // MFMAs per row tile
    constexpr int TOTAL_M = NT * MT;
    // 8 epilogue pieces of the pending pair, then 8 of this chunk's first pair, spread
    // over the MFMAs that exist (KS is 1 for b0, so pieces may share an MFMA there)
    // a pending pair of the PREVIOUS layer is this layer's k-step PQ, first read by MFMA
    // 2*PQ: all 8 pieces must have been issued before that
    constexpr bool PEND_EARLY = TOTAL_M < 24 || (PB != B && PB >= 0 && 2 * PQ < 12);
    constexpr int PEND_PER = PEND_EARLY ? 2 : 1;
    constexpr int PEND_M0 = PEND_EARLY ? 0 : 2;
    static_assert(PB < 0 || PB == B || 2 * PQ >= PEND_M0 + 8 / PEND_PER, "pending pair finished too late");
    constexpr int PAIR_M0 = TOTAL_M >= 24 ? 2 * MT + 2 : 2 * MT;
    constexpr int PAIR_PER = (TOTAL_M - PAIR_M0) >= 8 ? 1 : 2;
    const unsigned wb = c.b_wread[CC & 1];
    static_assert(bwd_mt(B) % TPC == 0, "row tiles per backward layer");

    Stage<CC>::issue(c);
    __builtin_amdgcn_sched_barrier(0);   // every other vector-memory instruction of the chunk stays behind the DMA

    ex8 a[AHEAD];
#pragma unroll
    for (int f = 0; f < AHEAD && f < F; ++f) a[f] = lds_load<ex8>(wb, f * 1024);
    // The next backward layer's ReLU mask is fetched a layer ahead, in this layer's SECOND chunk:
    // mk[(B+1) & 1] also held layer B-1's mask, whose last (pending) pair is finished in chunk 0.
    if constexpr (C == 1 && B + 1 < NUM_BWD) {
        if constexpr (bwd_desc(B + 1).mask_act >= 0) load_mask<bwd_desc(B + 1).mask_act>(st, st.mk[(B + 1) & 1]);
    }
    static_assert(layer_chunks(B) >= 2, "the mask prefetch lives in chunk 1");
    f32x4 acc[2][NT];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[cb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);

    // register lifetimes against the MFMA write-after-read hazard nops, as in mlp_bf16_16.hip chunk_step
    ex8 as_prev = a[0];
    f32x4 c_prev = acc[0][0];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int f = t * KS + ks;
            const ex8 as = a[f % AHEAD];
            if (f + AHEAD < F) a[f % AHEAD] = lds_load<ex8>(wb, (f + AHEAD) * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int m = f * 2 + cb;
                ex8 bs;
                if (ks < KS_CHAIN) bs = in[cb][ks < KS_CHAIN ? ks : 0];
                else bs = (B == 0) ? st.bx_rgb[cb] : st.bx_sig[cb];
                const f32x4 c_old = acc[cb][t];
                acc[cb][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as, bs, c_old, 0, 0, 0);
                asm volatile("" :: "v"(c_prev));
                c_prev = c_old;
                if constexpr (PB >= 0) {
                    if (m >= PEND_M0 && m < PEND_M0 + 8 / PEND_PER) {
#pragma unroll
                        for (int k = 0; k < PEND_PER; ++k) {
                            const int i = (m - PEND_M0) * PEND_PER + k;
                            if constexpr (PB == B) epilogue_piece<PB, PQ, F8>(i, st.pend, out, st);
                            else epilogue_piece<PB, PQ, F8>(i, st.pend, in, st);
                        }
                    }
                }
                if (m >= PAIR_M0 && m < PAIR_M0 + 8 / PAIR_PER) {
                    const f32x4 pr[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
#pragma unroll
                    for (int k = 0; k < PAIR_PER; ++k)
                        epilogue_piece<B, 2 * C, F8>((m - PAIR_M0) * PAIR_PER + k, pr, out, st);
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
    for (int cb = 0; cb < 2; ++cb) {
        st.pend[cb][0] = acc[cb][NT - 2];
        st.pend[cb][1] = acc[cb][NT - 1];
    }
    // this wave's LDS-DMA pieces (issued first in this chunk) have landed; what follows them may fly on:
    // 2 dY stores per finished pair (pending + this chunk's own) and the 4 mask loads of chunk 1
    constexpr bool PREFETCH = C == 1 && B + 1 < NUM_BWD && bwd_desc(B + 1 < NUM_BWD ? B + 1 : B).mask_act >= 0;
    // (8-bit form: a group of four fragments goes out with its last pair -- always an odd one, i.e. a pending pair: 4 data
    // stores + 1 exponent dword; nothing otherwise)
    constexpr int STORES = F8 ? ((PB >= 0 && (PQ & 3) == 3) ? F8_GROUP + 1 : 0) : (PB >= 0 ? 2 : 0) + 2;
    static_assert(!F8 || ((2 * C) & 3) != 3, "an in-chunk pair never ends a group");
    chunk_barrier<STORES + (PREFETCH ? 4 : 0)>();
}

__host__ __device__ constexpr int prev_layer(int b, int C) { return C > 0 ? b : b - 1; }
__host__ __device__ constexpr int prev_pair(int b, int C) {
    return C > 0 ? 2 * C - 1 : (b > 0 ? bwd_mt(b - 1) / 2 - 1 : 0);
}
template <int B, bool F8, int... Cs>
__device__ __forceinline__ void run_layer_seq(const Ctx& c, State& st, ex8 (&in)[2][8], ex8 (&out)[2][8],
                                              std::integer_sequence<int, Cs...>) {
    (chunk_step<B, Cs, prev_layer(B, Cs), prev_pair(B, Cs), F8>(c, st, in, out), ...);
}
template <int B, bool F8>
__device__ __forceinline__ void run_layer(const Ctx& c, State& st, ex8 (&in)[2][8], ex8 (&out)[2][8]) {
    run_layer_seq<B, F8>(c, st, in, out, std::make_integer_sequence<int, layer_chunks(B)>{});
}

template <bool F8>
__device__ __forceinline__ void bwd_body(const BwdArgs& a, long long ntiles) {
    Ctx c;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.image), 0, (int)BWD_IMAGE_BYTES, 0x00020000);
    c.wave_goff = c.wave * 1024;
    c.lane16 = c.lane * 16;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        c.b_wread[p] = LDS_W0 + p * LDS_WBUF + c.lane * 16;
        c.s_wdst[p] = LDS_W0 + p * LDS_WBUF + c.wave * 1024;
    }
    Stage<NUM_CHUNKS - 1>::issue(c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
    __syncthreads();

    const int col = c.lane & 15, g = c.lane >> 4;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long tile_base = tile * TILE_PTS;
        asm volatile("" : "+s"(c.wave_goff));
        State st;
        st.acts = a.acts;
        st.dys = a.dys;
        st.P = a.P;
        st.mask_tile = F8 ? f8_mask_offset_bytes(0, tile, 0, a.P) : mask_offset_bytes(0, tile, 0, a.P);
        st.tile = tile;
        load_mask<bwd_desc(0).mask_act>(st, st.mk[0]);
        const __bf16 z = (__bf16)0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const long long p = tile_base + c.wave * 32 + cb * 16 + col;
            st.loff[cb] = p < a.P ? block_lane_offset(g, c.wave * 32 + cb * 16 + col) : LOFF_INVALID;
            if (F8 && cb == 0)      // the lane writes the granule of point 16 (lane >> 5) + col of the wave (store_fragment_f8)
                st.loff[0] = tile_base + c.wave * 32 + 16 * (c.lane >> 5) + col < a.P ? f8_lane_offset(c.lane, c.wave) : LOFF_INVALID;
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            if (p < a.P) d = *reinterpret_cast<const f32x4*>(a.d_raw + p * 4);
            // custom k-steps: lane group 0 carries drgb (elements 0..2) / dsigma (element 0)
            const bool g0 = g == 0;
            st.bx_rgb[cb] = ex8{g0 ? (__bf16)d[0] : z, g0 ? (__bf16)d[1] : z, g0 ? (__bf16)d[2] : z, z, z, z, z, z};
            st.bx_sig[cb] = ex8{g0 ? (__bf16)d[3] : z, z, z, z, z, z, z, z};
        }
        run_layer<0, F8>(c, st, st.X, st.X);     // d c    (b0 reads only drgb)        -> X[.][0..3]
        run_layer<1, F8>(c, st, st.X, st.Y);     // d h9
        run_layer<2, F8>(c, st, st.Y, st.X);     // d h8
        run_layer<3, F8>(c, st, st.X, st.Y);     // d h7
        run_layer<4, F8>(c, st, st.Y, st.X);     // d h6
        run_layer<5, F8>(c, st, st.X, st.Y);     // d h5
        run_layer<6, F8>(c, st, st.Y, st.X);     // d h4
        run_layer<7, F8>(c, st, st.X, st.Y);     // d h3
        run_layer<8, F8>(c, st, st.Y, st.X);     // d h2
        run_layer<9, F8>(c, st, st.X, st.Y);     // d h1
        // the last pair of d h1 is still pending
#pragma unroll
        for (int i = 0; i < 8; ++i) epilogue_piece<9, bwd_mt(9) / 2 - 1, F8>(i, st.pend, st.Y, st);
    }
}

__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void nerf_mlp_bwd_kernel(BwdArgs a, long long ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    (void)smem;
    bwd_body<false>(a, ntiles);
}
// the same chain writing dY in the 8-bit storage form (and reading the masks behind an 8-bit activation buffer)
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void nerf_mlp_bwd_e4m3_kernel(BwdArgs a, long long ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    (void)smem;
    bwd_body<true>(a, ntiles);
}

}  // namespace

extern "C" int nerf_amd_launch_mlp_backward(const float* d_raw, const void* image, const void* acts, void* dys,
                                            long long P, int store_e4m3, hipStream_t stream) {
    (void)hipGetLastError();
    if (P <= 0) return 0;
    BwdArgs a{d_raw, image, reinterpret_cast<const char*>(acts), reinterpret_cast<char*>(dys), P, store_e4m3};
    const long long ntiles = (P + TILE_PTS - 1) / TILE_PTS;
    const int cus = device_cus();
    const long long grid = ntiles < cus ? ntiles : cus;
    auto kern = store_e4m3 ? nerf_mlp_bwd_e4m3_kernel : nerf_mlp_bwd_kernel;
    const hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_TOTAL);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * 64), LDS_TOTAL, stream, a, ntiles);
    return (int)hipGetLastError();
}
