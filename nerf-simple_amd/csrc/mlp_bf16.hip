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
//     one 32-point column block; the partner wave's MFMAs cover a wave's
//     epilogue (bias/ReLU/bf16 pack) and LDS latency.
//   * weights are streamed L2 -> LDS in chunks of one 32-row output tile
//     (4..20 KiB, double buffered, one barrier per chunk), shared by the 4 waves.
//   * the encoded position/direction fragments ("per-sample features") are
//     staged once per tile in LDS and read back by L0, the skip layer and the
//     colour layer.
//   * persistent grid: a workgroup walks tiles blockIdx.x, +gridDim.x, ...; the
//     chunk sequence is cyclic, so the next tile's first chunk is prefetched
//     during the last chunk of the current one.
#include "nerf_device.h"
#include <utility>

using namespace nerf_layout;

namespace {

constexpr int NB = 1;                         // 32-point column blocks per wave
constexpr int WAVES = 8;                      // waves per workgroup (2 per SIMD)
constexpr int TILE_PTS = WAVES * NB * 32;     // points per workgroup tile

// LDS carve-up (bytes); one dynamic array (cdna_hip_programming.md G17).
// Every LDS access below is (one of five per-lane base registers) + a 16-bit
// immediate, so no per-chunk address ever needs a register of its own.
constexpr int LDS_WBUF = 24 * 1024;           // one weight buffer (20 KiB chunk + staging slack)
constexpr int LDS_BIAS = 0;                                   // BIAS_FLOATS f32, padded to 10 KiB
constexpr int LDS_W0 = 10 * 1024;                             // 2 weight buffers
constexpr int LDS_POSD = LDS_W0 + 2 * LDS_WBUF;               // [wave][blk][2][1 KiB]
constexpr int LDS_POSX = LDS_POSD + WAVES * NB * 2 * 1024;    // [wave][blk][4][1 KiB]
constexpr int LDS_TOTAL = LDS_POSX + WAVES * NB * 4 * 1024;
static_assert(BIAS_FLOATS * 4 <= LDS_W0, "bias table");
static_assert(LDS_W0 + 2 * LDS_WBUF <= 65536, "weight reads must fit the ds offset field");
static_assert(LDS_TOTAL % 16 == 0 && LDS_TOTAL <= 160 * 1024, "LDS budget");
static_assert(NUM_CHUNKS % 2 == 0, "buffer parity must repeat per tile");
static_assert(BF16_MAX_CHUNK_KIB <= 24 && 24 % WAVES == 0, "staging geometry");

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

// the chunk sequence: chunk id -> (layer, m)
__host__ __device__ constexpr int chunk_layer(int cc) {
    int L = 0;
    while (cc >= layer_desc(L).mt) { cc -= layer_desc(L).mt; ++L; }
    return L;
}
__host__ __device__ constexpr int chunk_first(int L) {
    int c = 0;
    for (int i = 0; i < L; ++i) c += layer_desc(i).mt;
    return c;
}
__host__ __device__ constexpr int chunk_off_kib(int cc) {
    const int L = chunk_layer(cc);
    return bf16_layer_off_kib(L) + (cc - chunk_first(L)) * bf16_chunk_kib(L);
}
__host__ __device__ constexpr int chunk_kib(int cc) { return bf16_chunk_kib(chunk_layer(cc)); }

struct Ctx {
    __amdgpu_buffer_rsrc_t wrsrc;   // packed weight image (bounds-checked buffer)
    unsigned wave_goff;             // wave * 1024: this wave's piece inside a staging row (SGPR)
    unsigned lane16;                // lane * 16
    unsigned b_wread;               // LDS base for weight fragment reads   = lane*16
    unsigned b_wstore;              // LDS base for staging stores           = wave*1024 + lane*16
    unsigned b_bias;                // LDS base for bias reads               = (lane>>5)*16
    unsigned b_posx;                // LDS base of this wave's posx fragments
    unsigned b_posd;                // LDS base of this wave's posd fragments
    int wave, lane;
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
template <bool RELU>
__device__ __forceinline__ bf16x8 pack8(const f32x16& acc, int base) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack2<RELU>(acc[base + 2 * i], acc[base + 2 * i + 1]);
    return __builtin_bit_cast(bf16x8, r);
}

// ---- weight chunk staging: global -> registers -> LDS (double buffered) -----
// Every wave moves the same number of 1 KiB pieces (24 KiB / 8 waves = 3 for
// the big chunks); pieces past the chunk's end are junk that lands in the
// buffer's slack and is never read (the buffer descriptor bounds the reads).
template <int CC>
struct Stage {
    static constexpr int NEXT = (CC + 1) % NUM_CHUNKS;
    static constexpr int PIECES = (chunk_kib(NEXT) + WAVES - 1) / WAVES;
    static constexpr int SRC_OFF = chunk_off_kib(NEXT) * 1024;
    u32x4 r[PIECES];
    __device__ __forceinline__ void load(const Ctx& c) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            r[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                c.wrsrc, c.lane16, c.wave_goff + (SRC_OFF + p * WAVES * 1024), 0));
    }
    __device__ __forceinline__ void store(const Ctx& c) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            lds_store<u32x4>(c.b_wstore, LDS_W0 + (NEXT & 1) * LDS_WBUF + p * (WAVES * 1024), r[p]);
    }
};

// ---- one chunk = one 32-row output tile M of layer L ---------------------------
// in:  chain fragments (the previous layer's output); extra fragments (posx /
//      posd) come from this wave's LDS staging area
// out: this layer's output as the next layer's chain fragments
template <int L, int M>
__device__ __forceinline__ void chunk_step(const Ctx& c, const bf16x8 (&in)[NB][16],
                                           bf16x8 (&out)[NB][16], float (&sigma)[NB],
                                           float (&rgb)[NB][3]) {
    constexpr LayerDesc D = layer_desc(L);
    constexpr int KS_CHAIN = D.chain_k / 16;
    constexpr int KS_EXTRA = D.extra_slots / 16;
    constexpr int KS = KS_CHAIN + KS_EXTRA;
    constexpr int CC = chunk_first(L) + M;          // position in the cyclic chunk sequence
    constexpr int WB = LDS_W0 + (CC & 1) * LDS_WBUF;
    constexpr int EXTRA_BLK = (D.extra_kind == 1 ? 4 : 2) * 1024;
    constexpr int AHEAD = 4;                        // weight fragments in flight from LDS
    constexpr int BIAS_OFF = LDS_BIAS + (bias_off(L) + 32 * M) * 4;

    // stage the NEXT chunk (global -> registers now, registers -> LDS after the MFMAs)
    Stage<CC> st;
    st.load(c);

    // accumulators start from the bias of their rows:
    // register r of lane half h is row 32M + (r&3) + 8(r>>2) + 4h
    f32x16 acc[NB];
    {
        f32x16 binit;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = lds_load<f32x4>(c.b_bias, BIAS_OFF + 32 * g);
            binit[4 * g + 0] = b4[0]; binit[4 * g + 1] = b4[1];
            binit[4 * g + 2] = b4[2]; binit[4 * g + 3] = b4[3];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = binit;
    }
    bf16x8 a[AHEAD];
#pragma unroll
    for (int s = 0; s < AHEAD && s < KS; ++s) a[s] = lds_load<bf16x8>(c.b_wread, WB + s * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        // source order = issue order: the fragment AHEAD steps on is requested
        // before this step's MFMA (sched_barrier pins it; hipcc otherwise sinks
        // the read next to its use and every MFMA waits a full LDS round trip)
        const bf16x8 as = a[s % AHEAD];
        if (s + AHEAD < KS) a[s % AHEAD] = lds_load<bf16x8>(c.b_wread, WB + (s + AHEAD) * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            bf16x8 bf;
            if (s < KS_CHAIN) {
                bf = in[b][s < KS_CHAIN ? s : 0];
            } else {
                const unsigned xb = D.extra_kind == 1 ? c.b_posx : c.b_posd;
                bf = lds_load<bf16x8>(xb, b * EXTRA_BLK + (s - KS_CHAIN) * 1024);
            }
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as, bf, acc[b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if constexpr (L == 10) {
            rgb[b][0] = acc[b][0]; rgb[b][1] = acc[b][1]; rgb[b][2] = acc[b][2];
        } else if constexpr (L == 8 && M == 8) {
            sigma[b] = acc[b][0];
        } else {
            out[b][2 * M] = pack8<D.relu != 0>(acc[b], 0);
            out[b][2 * M + 1] = pack8<D.relu != 0>(acc[b], 8);
        }
    }
    st.store(c);
    __syncthreads();
}

template <int L, int... Ms>
__device__ __forceinline__ void run_layer_seq(const Ctx& c, const bf16x8 (&in)[NB][16],
                                              bf16x8 (&out)[NB][16], float (&sigma)[NB],
                                              float (&rgb)[NB][3], std::integer_sequence<int, Ms...>) {
    (chunk_step<L, Ms>(c, in, out, sigma, rgb), ...);
}
template <int L>
__device__ __forceinline__ void run_layer(const Ctx& c, const bf16x8 (&in)[NB][16],
                                          bf16x8 (&out)[NB][16], float (&sigma)[NB],
                                          float (&rgb)[NB][3]) {
    run_layer_seq<L>(c, in, out, sigma, rgb, std::make_integer_sequence<int, layer_desc(L).mt>{});
}

// ---- per-tile input stage: sample, encode, write B fragments to LDS ----------
template <bool RAYS>
__device__ __forceinline__ void stage_inputs(const Ctx& c, const MlpArgs& a, long long tile_base) {
    const int col = c.lane & 31, h = c.lane >> 5;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        long long p = tile_base + (c.wave * NB + b) * 32 + col;
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
                lds_store<u32x4>(c.b_posx, b * 4096 + s * 1024, r);
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
                lds_store<u32x4>(c.b_posd, b * 2048 + s * 1024, r);
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
    c.b_wread = c.lane * 16;
    c.b_wstore = c.wave * 1024 + c.lane * 16;
    c.b_bias = (c.lane >> 5) * 16;
    c.b_posx = LDS_POSX + c.wave * (NB * 4096) + c.lane * 16;
    c.b_posd = LDS_POSD + c.wave * (NB * 2048) + c.lane * 16;

    // prologue: bias table and chunk 0 -> LDS
    {
        const float* bsrc = reinterpret_cast<const float*>(
            reinterpret_cast<const char*>(a.packed) + (long long)BF16_WEIGHT_KIB * 1024);
        for (int i = threadIdx.x; i < BIAS_FLOATS; i += WAVES * 64)
            lds_store<float>(i * 4, LDS_BIAS, bsrc[i]);
        Stage<NUM_CHUNKS - 1> st;          // NEXT == chunk 0
        st.load(c);
        st.store(c);
    }
    __syncthreads();

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long tile_base = tile * TILE_PTS;
        // keep per-chunk scalar offsets (wave_goff + constant) from being
        // hoisted out of the tile loop and spilled: make the base loop-variant
        asm volatile("" : "+s"(c.wave_goff));
        stage_inputs<RAYS>(c, a, tile_base);

        bf16x8 A[NB][16], B[NB][16];
        float sigma[NB], rgb[NB][3];
        run_layer<0>(c, A, A, sigma, rgb);      // L0 reads only the posx fragments
        run_layer<1>(c, A, B, sigma, rgb);
        run_layer<2>(c, B, A, sigma, rgb);
        run_layer<3>(c, A, B, sigma, rgb);
        run_layer<4>(c, B, A, sigma, rgb);
        run_layer<5>(c, A, B, sigma, rgb);
        run_layer<6>(c, B, A, sigma, rgb);
        run_layer<7>(c, A, B, sigma, rgb);
        run_layer<8>(c, B, A, sigma, rgb);
        run_layer<9>(c, A, B, sigma, rgb);
        run_layer<10>(c, B, A, sigma, rgb);

        // rows 0..2 (rgb) and row 256 (sigma) live in registers 0..2 / 0 of
        // lane half 0; column = point
        if (c.lane < 32) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const long long p = tile_base + (c.wave * NB + b) * 32 + c.lane;
                if (p < a.P) {
                    const f32x4 o = {rgb[b][0], rgb[b][1], rgb[b][2], sigma[b]};
                    *reinterpret_cast<f32x4*>(a.raw + p * 4) = o;
                }
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
