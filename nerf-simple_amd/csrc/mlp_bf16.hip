// mlp_bf16.hip -- fused sampling + positional encoding + 12-layer MLP for
// gfx950, bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// Replaces, for one tile of 256 query points per workgroup, the reference's
//   utils/rendering.py:24-40 (stratified sampling, point assembly),
//   utils/xyz.py:6-36       (positional encoding),
//   utils/nets.py:34-43     (Nerf.forward)
// and writes raw[P,4] = [r,g,b,sigma] (+ ts[B,N] in rays mode).
//
// Design:
//   * H^T = W . X^T: output features on MFMA rows, points on MFMA columns, so
//     each layer's accumulator tile, converted to bf16 in registers, IS the
//     next layer's B operand -- hidden activations never touch LDS or HBM.
//   * one workgroup = 8 waves (two per SIMD, <= 256 registers each); a wave owns
//     one 32-point column block (64 VGPRs of input features, 64 of output).
//   * weights stream L2 -> LDS by LDS-DMA in chunks of TWO 32-row output tiles
//     (8..40 KiB), double buffered, ONE barrier per chunk, shared by the 8 waves.
//   * the instruction stream of a chunk is laid out by hand (sched_barrier
//     pins it): weight fragments are requested 4 MFMAs ahead, the bias/ReLU/bf16
//     pack of a finished tile is issued two ops at a time under the NEXT tile's
//     MFMAs -- also across chunk and layer boundaries (the "pending" tile).
//   * the encoded position/direction fragments ("per-sample features") are
//     staged once per tile in LDS and read back (prefetched) by L0, the skip
//     layer and the colour layer.
//   * persistent grid: a workgroup walks tiles blockIdx.x, +gridDim.x, ...; the
//     chunk sequence is cyclic, so the next tile's first chunk is prefetched
//     during the last chunk of the current one.
#include "nerf_device.h"
#include <utility>

using namespace nerf_layout;

namespace {

constexpr int WAVES = 8;                      // waves per workgroup (2 per SIMD)
constexpr int TILE_PTS = WAVES * 32;          // points per workgroup tile
constexpr int TPC = 2;                        // 32-row output tiles per weight chunk

// ---- the chunk sequence ------------------------------------------------------
__host__ __device__ constexpr int layer_chunks(int L) { return (layer_desc(L).mt + TPC - 1) / TPC; }
__host__ __device__ constexpr int chunk_first(int L) {
    int c = 0;
    for (int i = 0; i < L; ++i) c += layer_chunks(i);
    return c;
}
constexpr int NUM_CHUNKS = chunk_first(NUM_LAYERS);           // 40
__host__ __device__ constexpr int chunk_layer(int cc) {
    int L = 0;
    while (cc >= layer_chunks(L)) { cc -= layer_chunks(L); ++L; }
    return L;
}
__host__ __device__ constexpr int chunk_tiles(int cc) {      // tiles in chunk cc
    const int L = chunk_layer(cc), C = cc - chunk_first(L);
    const int left = layer_desc(L).mt - C * TPC;
    return left < TPC ? left : TPC;
}
__host__ __device__ constexpr int chunk_kib(int cc) { return chunk_tiles(cc) * bf16_chunk_kib(chunk_layer(cc)); }
__host__ __device__ constexpr int chunk_off_kib(int cc) {
    const int L = chunk_layer(cc), C = cc - chunk_first(L);
    return bf16_layer_off_kib(L) + C * TPC * bf16_chunk_kib(L);
}

// ---- LDS carve-up (bytes); one dynamic array (cdna_hip_programming.md G17) ----
// Every LDS access is (a per-lane base register) + a 16-bit immediate, so no
// per-chunk address ever needs a register of its own.
constexpr int LDS_WBUF = 40 * 1024;           // one weight buffer = the largest chunk
constexpr int LDS_BIAS = 0;                                   // BIAS_FLOATS f32, padded to 10 KiB
constexpr int LDS_W0 = 10 * 1024;                             // 2 weight buffers
constexpr int LDS_POSD = LDS_W0 + 2 * LDS_WBUF;               // [wave][2][1 KiB]
constexpr int LDS_POSX = LDS_POSD + WAVES * 2 * 1024;         // [wave][4][1 KiB]
constexpr int LDS_TOTAL = LDS_POSX + WAVES * 4 * 1024;
static_assert(BIAS_FLOATS * 4 <= LDS_W0, "bias table");
static_assert(LDS_TOTAL % 16 == 0 && LDS_TOTAL <= 160 * 1024, "LDS budget");
static_assert(NUM_CHUNKS % 2 == 0, "buffer parity must repeat per tile");
static_assert(TPC * BF16_MAX_CHUNK_KIB * 1024 <= LDS_WBUF && LDS_WBUF % (WAVES * 1024) == 0, "staging geometry");

typedef __attribute__((address_space(3))) char lds_char;
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
    __amdgpu_buffer_rsrc_t wrsrc;   // packed weight image (bounds-checked buffer)
    unsigned wave_goff;             // wave * 1024: this wave's piece inside a staging row (SGPR)
    unsigned lane16;                // lane * 16
    unsigned b_wread[2];            // LDS base of weight buffer p for fragment reads  (+ lane*16)
    unsigned s_wdst[2];             // LDS address of this wave's piece in weight buffer p (wave-uniform)
    unsigned b_bias;                // LDS base for bias reads = (lane>>5)*16
    unsigned b_posx, b_posd;        // LDS base of this wave's posx / posd fragments
    int wave, lane;
};

// All per-point state a wave carries through the network.
struct State {
    bf16x8 X[16], Y[16];            // ping-pong activation fragments (64 VGPRs each)
    f32x16 pend;                    // accumulator tile whose epilogue is still owed
    float sigma, rgb[3];
};

// relu + f32 -> bf16 for a register pair: v_cvt_pk_bf16_f32 + v_pk_max_i16
// (a negative bf16 is a negative int16, so integer max with 0 is ReLU)
template <bool RELU>
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    if constexpr (RELU) {
        const s16x2 z = {0, 0};
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, r), z));
    } else {
        return __builtin_bit_cast(unsigned, r);
    }
}

// ---- weight chunk staging: L2 -> LDS by LDS-DMA (buffer_load ... lds) ---------
// No VGPRs and no ds_write: one wave-instruction moves 1 KiB (64 lanes x 16 B)
// to a wave-uniform LDS address.  Every wave moves the same number of pieces;
// pieces past the chunk's end are junk that lands in the buffer's slack and is
// never read (the buffer descriptor bounds the global reads).  The DMA is
// issued at the start of a chunk and retired by the vmcnt(0) of the barrier
// that ends it, a whole chunk of MFMAs later.
typedef __attribute__((address_space(3))) void lds_void;
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

// ---- the epilogue of one finished tile, in 8 independent pieces ----------------
// Tile M of layer L: accumulator registers 2i, 2i+1 -> word i&3 of fragment
// 2M + (i>>2) of the layer's output array (ReLU + bf16), or the sigma / rgb
// extraction for the two head tiles.
template <int L, int M>
__device__ __forceinline__ void epilogue_piece(int i, const f32x16& acc, bf16x8 (&dst)[16], State& st) {
    constexpr LayerDesc D = layer_desc(L);
    if constexpr (L == 10) {
        if (i == 0) { st.rgb[0] = acc[0]; st.rgb[1] = acc[1]; st.rgb[2] = acc[2]; }
    } else if constexpr (L == 8 && M == 8) {
        if (i == 0) st.sigma = acc[0];
    } else {
        u32x4 w = __builtin_bit_cast(u32x4, dst[2 * M + (i >> 2)]);
        w[i & 3] = pack2<D.relu != 0>(acc[2 * i], acc[2 * i + 1]);
        dst[2 * M + (i >> 2)] = __builtin_bit_cast(bf16x8, w);
    }
}

__device__ __forceinline__ void load_bias(const Ctx& c, int off, f32x16& acc) {
    // register r of lane half h is row 32M + (r&3) + 8(r>>2) + 4h
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = lds_load<f32x4>(c.b_bias, off + 32 * g);
        acc[4 * g + 0] = b4[0]; acc[4 * g + 1] = b4[1];
        acc[4 * g + 2] = b4[2]; acc[4 * g + 3] = b4[3];
    }
}

// ---- one chunk: NT output tiles of layer L, starting at tile M0 = C*TPC -------
// PL/PM: layer / tile of the pending accumulator handed over by the previous
// chunk (PL < 0: none).  in = this layer's input fragments, out = its output.
template <int L, int C, int PL, int PM>
__device__ __forceinline__ void chunk_step(const Ctx& c, State& st, bf16x8 (&in)[16], bf16x8 (&out)[16]) {
    constexpr LayerDesc D = layer_desc(L);
    constexpr int KS_CHAIN = D.chain_k / 16;
    constexpr int KS_EXTRA = D.extra_slots / 16;
    constexpr int KS = KS_CHAIN + KS_EXTRA;
    constexpr int CC = chunk_first(L) + C;          // position in the cyclic chunk sequence
    constexpr int NT = chunk_tiles(CC);
    constexpr int M0 = C * TPC;
    constexpr int TILE_BYTES = bf16_chunk_kib(L) * 1024;
    constexpr int F = NT * KS;                      // MFMAs in this chunk
    constexpr int AHEAD = 4;                        // weight fragments in flight from LDS
    constexpr int XAHEAD = 2;                       // extra (posx/posd) fragments in flight
    constexpr int BIAS_OFF = LDS_BIAS + (bias_off(L) + 32 * M0) * 4;
    constexpr int S0 = KS > 4 ? 2 : 0;              // first k-step that carries epilogue pieces
    using St = Stage<CC>;
    const unsigned wb = c.b_wread[CC & 1];
    const unsigned xb = D.extra_kind == 1 ? c.b_posx : c.b_posd;

    St::issue(c);

    // chunk-linear step index f = t*KS + s
    auto w_addr = [](int f) { return (f / KS) * TILE_BYTES + (f % KS) * 1024; };
    auto is_extra = [](int f) { return f < F && (f % KS) >= KS_CHAIN; };
    auto extra_e = [](int f) { return (f % KS) - KS_CHAIN; };                  // which extra fragment
    auto extra_j = [](int f) { return (f / KS) * KS_EXTRA + (f % KS) - KS_CHAIN; };   // its use counter

    bf16x8 a[AHEAD], bx[XAHEAD];
#pragma unroll
    for (int f = 0; f < AHEAD && f < F; ++f) a[f] = lds_load<bf16x8>(wb, w_addr(f));
    if constexpr (KS_EXTRA > 0) {
#pragma unroll
        for (int f = 0; f < XAHEAD; ++f)
            if (is_extra(f)) bx[extra_j(f) % XAHEAD] = lds_load<bf16x8>(xb, extra_e(f) * 1024);
    }
    f32x16 acc[NT];
    load_bias(c, BIAS_OFF, acc[0]);
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int f = t * KS + s;
            // ---- operands of this step, then the requests issued ahead of its MFMA
            //      (source order = issue order; hipcc would sink each read to its use)
            const bf16x8 as = a[f % AHEAD];
            bf16x8 bs;
            if (s < KS_CHAIN) bs = in[s < KS_CHAIN ? s : 0];
            else bs = bx[(KS_EXTRA > 0 ? extra_j(f) : 0) % XAHEAD];
            if (f + AHEAD < F) a[f % AHEAD] = lds_load<bf16x8>(wb, w_addr(f + AHEAD));
            if constexpr (KS_EXTRA > 0) {
                if (is_extra(f + XAHEAD))
                    bx[extra_j(f + XAHEAD) % XAHEAD] = lds_load<bf16x8>(xb, extra_e(f + XAHEAD) * 1024);
            }
            if (NT > 1 && t == 0 && s == KS / 2) load_bias(c, BIAS_OFF + 128, acc[NT - 1]);
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as, bs, acc[t], 0, 0, 0);
            // ---- work issued in this MFMA's shadow
            if (s >= S0 && s < S0 + 4) {
                const int i0 = 2 * (s - S0);
                if (t == 0) {
                    if constexpr (PL >= 0) {
                        // the previous chunk's last tile: same layer -> out, previous layer -> in
                        if constexpr (PL == L) {
                            epilogue_piece<PL, PM>(i0, st.pend, out, st);
                            epilogue_piece<PL, PM>(i0 + 1, st.pend, out, st);
                        } else {
                            epilogue_piece<PL, PM>(i0, st.pend, in, st);
                            epilogue_piece<PL, PM>(i0 + 1, st.pend, in, st);
                        }
                    }
                } else {
                    epilogue_piece<L, M0>(i0, acc[0], out, st);
                    epilogue_piece<L, M0>(i0 + 1, acc[0], out, st);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    st.pend = acc[NT - 1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
    __syncthreads();
}

// the tile that is pending when chunk (L, C) starts
__host__ __device__ constexpr int prev_layer(int L, int C) { return C > 0 ? L : L - 1; }
__host__ __device__ constexpr int prev_tile(int L, int C) {
    return C > 0 ? C * TPC - 1 : (L > 0 ? layer_desc(L - 1).mt - 1 : 0);
}

template <int L, int... Cs>
__device__ __forceinline__ void run_layer_seq(const Ctx& c, State& st, bf16x8 (&in)[16], bf16x8 (&out)[16],
                                              std::integer_sequence<int, Cs...>) {
    (chunk_step<L, Cs, prev_layer(L, Cs), prev_tile(L, Cs)>(c, st, in, out), ...);
}
template <int L>
__device__ __forceinline__ void run_layer(const Ctx& c, State& st, bf16x8 (&in)[16], bf16x8 (&out)[16]) {
    static_assert(TPC == 2, "the epilogue interleave assumes two tiles per chunk");
    run_layer_seq<L>(c, st, in, out, std::make_integer_sequence<int, layer_chunks(L)>{});
}

// ---- per-tile input stage: sample, encode, write B fragments to LDS ----------
template <bool RAYS>
__device__ __forceinline__ void stage_inputs(const Ctx& c, const MlpArgs& a, long long tile_base) {
    const int col = c.lane & 31, h = c.lane >> 5;
    {
        long long p = tile_base + c.wave * 32 + col;
        const bool valid = p < a.P;
        if (!valid) p = a.P - 1;
        PointIn pt;
        if constexpr (RAYS) {
            pt = fetch_point_rays(a, p);
            if (valid && h == 0 && a.ts_out) a.ts_out[p] = pt.t;
        } else {
            pt = fetch_point_pts(a, p);
        }
        // ---- posx: 32 slots per lane half (nerf_layout::posx_col) ----
        {
            const float xyz[3] = {pt.x, pt.y, pt.z};
            const float hs = h ? 32.f : 1.f;               // 2^(5h)
            float v[32];
#pragma unroll
            for (int cd = 0; cd < 3; ++cd) {
                TwoF q = to_revolutions(xyz[cd]);
                q.hi *= hs; q.lo *= hs;
#pragma unroll
                for (int lv = 0; lv < 5; ++lv) {
                    const int pp = cd * 5 + lv;
                    sincos_rev_fast(q, (float)(1 << lv), v[2 * pp], v[2 * pp + 1]);
                }
            }
            v[30] = h ? pt.z : pt.x;
            v[31] = h ? 0.f : pt.y;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4 r;
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = pack2<false>(v[8 * s + 2 * i], v[8 * s + 2 * i + 1]);
                lds_store<u32x4>(c.b_posx, s * 1024, r);
            }
        }
        // ---- posd: 16 slots per lane half (nerf_layout::posd_col) ----
        {
            const float dd[3] = {pt.d1, pt.d2, pt.d3};
            const float hs = h ? 4.f : 1.f;                // 2^(2h)
            float v[16];
#pragma unroll
            for (int cd = 0; cd < 3; ++cd) {
                TwoF q = to_revolutions(dd[cd]);
                q.hi *= hs; q.lo *= hs;
#pragma unroll
                for (int lv = 0; lv < 2; ++lv) {
                    const int pp = cd * 2 + lv;
                    sincos_rev_fast(q, (float)(1 << lv), v[2 * pp], v[2 * pp + 1]);
                }
            }
            v[12] = h ? pt.d3 : pt.d1;
            v[13] = h ? 0.f : pt.d2;
            v[14] = 0.f; v[15] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4 r;
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = pack2<false>(v[8 * s + 2 * i], v[8 * s + 2 * i + 1]);
                lds_store<u32x4>(c.b_posd, s * 1024, r);
            }
        }
    }
}

template <bool RAYS>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void nerf_mlp_bf16_kernel(MlpArgs a, long long ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    (void)smem;     // carved by absolute LDS offsets (the dynamic region starts at 0)
    Ctx c;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.packed), 0,
                                                (int)BF16_PACKED_BYTES, 0x00020000);
    c.wave_goff = c.wave * 1024;
    c.lane16 = c.lane * 16;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        c.b_wread[p] = LDS_W0 + p * LDS_WBUF + c.lane * 16;
        c.s_wdst[p] = LDS_W0 + p * LDS_WBUF + c.wave * 1024;
    }
    c.b_bias = (c.lane >> 5) * 16;
    c.b_posx = LDS_POSX + c.wave * 4096 + c.lane * 16;
    c.b_posd = LDS_POSD + c.wave * 2048 + c.lane * 16;

    // prologue: bias table and chunk 0 -> LDS
    {
        const float* bsrc = reinterpret_cast<const float*>(
            reinterpret_cast<const char*>(a.packed) + (long long)BF16_WEIGHT_KIB * 1024);
        for (int i = threadIdx.x; i < BIAS_FLOATS; i += WAVES * 64)
            lds_store<float>(i * 4, LDS_BIAS, bsrc[i]);
        Stage<NUM_CHUNKS - 1>::issue(c);         // NEXT == chunk 0
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
    __syncthreads();

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long tile_base = tile * TILE_PTS;
        // keep per-chunk scalar offsets (wave_goff + constant) from being
        // hoisted out of the tile loop and spilled: make the base loop-variant
        asm volatile("" : "+s"(c.wave_goff));
        stage_inputs<RAYS>(c, a, tile_base);

        State st;
        run_layer<0>(c, st, st.X, st.X);    // L0 reads only the posx fragments; its output goes to X
        run_layer<1>(c, st, st.X, st.Y);
        run_layer<2>(c, st, st.Y, st.X);
        run_layer<3>(c, st, st.X, st.Y);
        run_layer<4>(c, st, st.Y, st.X);
        run_layer<5>(c, st, st.X, st.Y);
        run_layer<6>(c, st, st.Y, st.X);
        run_layer<7>(c, st, st.X, st.Y);
        run_layer<8>(c, st, st.Y, st.X);
        run_layer<9>(c, st, st.X, st.Y);
        run_layer<10>(c, st, st.Y, st.X);
        // the colour head's tile is still pending: rows 0..2 = rgb
        epilogue_piece<10, 0>(0, st.pend, st.X, st);

        // rgb and sigma (row 256 of L8) live in registers 0..2 / 0 of lane half 0; column = point
        if (c.lane < 32) {
            const long long p = tile_base + c.wave * 32 + c.lane;
            if (p < a.P) {
                const f32x4 o = {st.rgb[0], st.rgb[1], st.rgb[2], st.sigma};
                *reinterpret_cast<f32x4*>(a.raw + p * 4) = o;
            }
        }
    }
}

}  // namespace

extern "C" int nerf_amd_launch_mlp_bf16(const MlpArgs* args, int rays_mode, hipStream_t stream) {
    (void)hipGetLastError();   // drop any stale error: the return value is about THIS launch
    MlpArgs a = *args;
    if (a.P <= 0) return 0;
    const long long ntiles = (a.P + TILE_PTS - 1) / TILE_PTS;
    int dev = 0, cus = 256;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    const long long grid = ntiles < cus ? ntiles : cus;
    auto kern = rays_mode ? nerf_mlp_bf16_kernel<true> : nerf_mlp_bf16_kernel<false>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * 64), LDS_TOTAL, stream, a, ntiles);
    return (int)hipGetLastError();
}
