// mlp_f32.hip -- fused sampling + positional encoding + 12-layer MLP for
// gfx950 in EXACT fp32: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate; each
// result is bit-for-bit an fmaf chain, cdna_hip_programming.md section 3).
// The fp32 configuration of BASELINE.json (config 2) and the tight-tolerance
// parity path against the CPU oracle.
//
// Same structure as mlp_bf16.hip (read its header first); differences:
//   * 16-row output tiles, 16 points per wave, 8 waves = 128 points per tile;
//   * an accumulator register is directly one k-step (4 features, one per lane
//     group) of the next layer -- no conversion at all between layers;
//   * the positional encoding uses the accurate ocml sinf/cosf on the exactly
//     scaled argument, so inputs track the torch-CPU encoder to ~1 ulp;
//   * COMP (the render path): compositing in the same launch out of an LDS ring of 1024 samples, as in
//     mlp_bf16_16.hip (a workgroup owns a contiguous range of rays; composite_device.h is the routine
//     composite.hip uses, so the pixels are bit-identical to the two-launch path).
// Peak for this path is the fp32 MFMA rate, 157.3 TFLOP/s (1/16 of bf16).
#include "composite_device.h"
#include <utility>

using namespace nerf_layout;

namespace {

constexpr int WAVES = 8;
constexpr int TILE_PTS = WAVES * 16;

constexpr int LDS_WBUF = 24 * 1024;
constexpr int LDS_BIAS = 0;
constexpr int LDS_W0 = 10 * 1024;
constexpr int LDS_POSD = LDS_W0 + 2 * LDS_WBUF;               // [wave][2][1 KiB]
constexpr int LDS_POSX = LDS_POSD + WAVES * 2 * 1024;         // [wave][4][1 KiB]
constexpr int LDS_TOTAL = LDS_POSX + WAVES * 4 * 1024;
// COMP only: the sample ring (16 B + 4 B per sample)
constexpr int RING_PTS = 1024;
constexpr int LDS_RING_RAW = LDS_TOTAL;
constexpr int LDS_RING_T = LDS_RING_RAW + RING_PTS * 16;
constexpr int LDS_TOTAL_COMP = LDS_RING_T + RING_PTS * 4;
static_assert(LDS_TOTAL_COMP <= 160 * 1024 && RING_PTS - TILE_PTS >= FUSED_RENDER_MAX_N, "ring: an unfinished ray plus a tile");
static_assert(F32_BIAS_FLOATS * 4 <= LDS_W0, "bias table");
static_assert(F32_NUM_CHUNKS % 2 == 0, "buffer parity must repeat per tile");

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

__host__ __device__ constexpr int chunk_layer(int cc) {
    int L = 0;
    while (cc >= f32_mt(L)) { cc -= f32_mt(L); ++L; }
    return L;
}
__host__ __device__ constexpr int chunk_first(int L) {
    int c = 0;
    for (int i = 0; i < L; ++i) c += f32_mt(i);
    return c;
}
__host__ __device__ constexpr int chunk_off_kib(int cc) {
    const int L = chunk_layer(cc);
    return f32_layer_off_kib(L) + (cc - chunk_first(L)) * f32_chunk_kib(L);
}
__host__ __device__ constexpr int chunk_kib(int cc) { return f32_chunk_kib(chunk_layer(cc)); }

struct Ctx {
    __amdgpu_buffer_rsrc_t wrsrc;
    unsigned wave_goff, lane16;
    unsigned b_wread, b_wstore, b_bias, b_posx, b_posd;
    unsigned posd_addr, posd_stride;   // this tile's direction fragment of the lane and the byte step between its two halves (stage_inputs)
    int wave, lane;
};

template <int CC>
struct Stage {
    static constexpr int NEXT = (CC + 1) % F32_NUM_CHUNKS;
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

// The first AHEAD weight fragment groups and the bias vector of the chunk that runs next, read right behind
// the barrier that publishes its buffer.  That barrier sits TAIL fragment groups BEFORE the end of a chunk
// (as in mlp_bf16_16.hip): the staged weights are written to LDS there, and the LDS round trip of the next
// chunk's first reads is covered by this chunk's last 4 * TAIL MFMAs instead of idling the matrix pipe at
// each of the 154 chunk starts of a tile.
struct WFrag {
    f32x4 a[2];
    f32x4 bias;
};

// one chunk = 16-row output tile T of layer L
template <int L, int T>
__device__ __forceinline__ void chunk_step(const Ctx& c, const float (&in)[64], float (&out)[64],
                                           float& sigma, float (&rgb)[3], WFrag& wf) {
    constexpr LayerDesc D = layer_desc(L);
    constexpr int Q_CHAIN = D.chain_k / 16;         // groups of 4 k-steps from the chain
    constexpr int Q_EXTRA = D.extra_slots / 16;
    constexpr int Q = Q_CHAIN + Q_EXTRA;
    constexpr int CC = chunk_first(L) + T;
    constexpr int WB = LDS_W0 + (CC & 1) * LDS_WBUF;
    constexpr int AHEAD = 2;
    // the chunk that runs next (cyclic over tiles)
    constexpr int NCC = (CC + 1) % F32_NUM_CHUNKS;
    constexpr int NL = chunk_layer(NCC);
    constexpr int NQ = layer_desc(NL).chain_k / 16 + layer_desc(NL).extra_slots / 16;
    constexpr int NWB = LDS_W0 + (NCC & 1) * LDS_WBUF;
    constexpr int NBIAS_OFF = LDS_BIAS + (f32_bias_off(NL) + 16 * (NCC - chunk_first(NL))) * 4;
    constexpr int TAIL = 2;
    constexpr int QB = Q > TAIL + AHEAD ? Q - TAIL : Q;       // barrier in front of group QB (Q: at the end)

    Stage<CC> st;
    st.load(c);
    auto publish_and_prefetch = [&]() {
        // every read of this chunk's buffer has been issued at least AHEAD groups ago; the other buffer was last
        // read in front of the previous chunk's barrier: it can take the next chunk's weights now
        st.store(c);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < AHEAD && q < NQ; ++q) wf.a[q] = lds_load<f32x4>(c.b_wread, NWB + q * 1024);
        wf.bias = lds_load<f32x4>(c.b_bias, NBIAS_OFF);
        __builtin_amdgcn_sched_barrier(0);
    };

    // register i of lane group g is row 16T + 4g + i
    f32x4 acc0 = wf.bias;
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 a[AHEAD];
#pragma unroll
    for (int q = 0; q < AHEAD && q < Q; ++q) a[q] = wf.a[q];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (q == QB) publish_and_prefetch();
        const f32x4 aq = a[q % AHEAD];
        if (q + AHEAD < Q) a[q % AHEAD] = lds_load<f32x4>(c.b_wread, WB + (q + AHEAD) * 1024);
        f32x4 bq;
        if (q < Q_CHAIN) {
            const int qq = q < Q_CHAIN ? q : 0;
            bq[0] = in[4 * qq + 0]; bq[1] = in[4 * qq + 1]; bq[2] = in[4 * qq + 2]; bq[3] = in[4 * qq + 3];
        } else {
            bq = D.extra_kind == 1 ? lds_load<f32x4>(c.b_posx, (q - Q_CHAIN) * 1024)
                                   : lds_load<f32x4>(c.posd_addr + (q - Q_CHAIN) * c.posd_stride, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // two accumulation chains (dependent-issue latency 40 > issue 32 cycles)
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[0], bq[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[1], bq[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[2], bq[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[3], bq[3], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const f32x4 r = acc0 + acc1;
    if constexpr (L == 10) {
        rgb[0] = r[0]; rgb[1] = r[1]; rgb[2] = r[2];
    } else if constexpr (L == 8 && T == 16) {
        sigma = r[0];
    } else {
#pragma unroll
        // ReLU as torch computes it: NaN stays NaN (fmaxf / v_max_f32 would return the 0), so that a non-finite input
        // or weight shows in the outputs exactly as it does in the reference (utils/nets.py:16-26, nn.ReLU)
        for (int i = 0; i < 4; ++i) out[4 * T + i] = D.relu ? (r[i] < 0.f ? 0.f : r[i]) : r[i];
    }
    if constexpr (QB == Q) publish_and_prefetch();
}

template <int L, int... Ts>
__device__ __forceinline__ void run_layer_seq(const Ctx& c, const float (&in)[64], float (&out)[64],
                                              float& sigma, float (&rgb)[3], WFrag& wf,
                                              std::integer_sequence<int, Ts...>) {
    (chunk_step<L, Ts>(c, in, out, sigma, rgb, wf), ...);
}
template <int L>
__device__ __forceinline__ void run_layer(const Ctx& c, const float (&in)[64], float (&out)[64],
                                          float& sigma, float (&rgb)[3], WFrag& wf) {
    run_layer_seq<L>(c, in, out, sigma, rgb, wf, std::make_integer_sequence<int, f32_mt(L)>{});
}

__device__ __forceinline__ float enc_exact(float x, int idx) {
    const float a = ldexpf(x, idx >> 1);            // 2^level * x, exact
    float s, co;
    sincosf(a, &s, &co);
    return (idx & 1) ? co : s;
}

// p_end: one past the last valid point (a.P, or the end of the workgroup's ray range in the fused render);
// ring_q0 >= 0: also drop the sample position into the ring slot of this point
template <bool RAYS>
__device__ __forceinline__ void stage_inputs(Ctx& c, const MlpArgs& a, long long tile_base, long long p_end,
                                             int ring_q0) {
    const int col = c.lane & 15, g = c.lane >> 4;
    long long p = tile_base + c.wave * 16 + col;
    const bool valid = p < p_end;
    if (!valid) p = p_end - 1;
    PointIn pt;
    c.posd_addr = c.b_posd;
    c.posd_stride = 1024;
    if constexpr (RAYS) {
        // Direction features once per RAY of the tile instead of once per sample (mlp_bf16_16.hip stage_inputs has the
        // reasoning): here each of them is an exact sincosf, a quarter of the tile's encoding work.  One LDS slot of
        // 128 B per ray, laid out [half q][lane group g][16 B]; value (g, slot s) by thread 32 * ray + 8 g + s.
        const long long b0 = tile_base / a.N;
        const long long last = (tile_base + TILE_PTS < p_end ? tile_base + TILE_PTS : p_end) - 1;
        const int nrays = (int)(last / a.N - b0) + 1;
        for (int idx = threadIdx.x; idx < nrays * 32; idx += WAVES * 64) {
            const int ray = idx >> 5, slot = idx & 31, gg = slot >> 3, sl = slot & 7;
            const float* rp = a.rays + (b0 + ray) * 6 + 3;
            const float dx = rp[0], dy = rp[1], dz = rp[2];
            const float nrm = norm3(dx, dy, dz);
            float val = 0.f;
            if (sl < 6) {
                const int cd = sl >> 1;
                val = enc_exact(__fdiv_rn(cd == 0 ? dx : cd == 1 ? dy : dz, nrm), 2 * gg + (sl & 1));
            } else if (sl == 6 && gg < 3) {
                val = __fdiv_rn(gg == 0 ? dx : gg == 1 ? dy : dz, nrm);
            }
            lds_store<float>(LDS_POSD + ray * 128 + (sl >> 2) * 64 + gg * 16 + (sl & 3) * 4, 0, val);
        }
        const RaySample rs = split_point(p, a.N);
        c.posd_addr = LDS_POSD + (unsigned)(rs.b - b0) * 128 + g * 16;
        c.posd_stride = 64;
        pt = fetch_point_rays<false>(a, p, rs);
        if (valid && g == 0 && a.ts_out) a.ts_out[p] = pt.t;
        if (ring_q0 >= 0 && valid && g == 0)
            lds_store<float>(((unsigned)(ring_q0 + c.wave * 16 + col) & (RING_PTS - 1)) * 4, LDS_RING_T, pt.t);
    } else {
        pt = fetch_point_pts(a, p);
    }
    // posx: 16 slots per lane group (nerf_layout::posx_col_f32)
    {
        const float xyz[3] = {pt.x, pt.y, pt.z};
        float v[16];
#pragma unroll
        for (int t = 0; t < 15; ++t) v[t] = enc_exact(xyz[t / 5], 5 * g + t % 5);
        v[15] = g == 0 ? pt.x : g == 1 ? pt.y : g == 2 ? pt.z : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 r = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            lds_store<f32x4>(c.b_posx, q * 1024, r);
        }
    }
    // posd: 8 slots per lane group (nerf_layout::posd_col_f32) -- per sample only when the directions are (points mode)
    if constexpr (!RAYS) {
        const float dd[3] = {pt.d1, pt.d2, pt.d3};
        float v[8];
#pragma unroll
        for (int t = 0; t < 6; ++t) v[t] = enc_exact(dd[t / 2], 2 * g + t % 2);
        v[6] = g == 0 ? pt.d1 : g == 1 ? pt.d2 : g == 2 ? pt.d3 : 0.f;
        v[7] = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 r = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            lds_store<f32x4>(c.b_posd, q * 1024, r);
        }
    }
}

struct RingSamples {                         // a ray's samples in the workgroup's LDS ring
    unsigned q0;
    __device__ __forceinline__ float t(int i) const {
        return lds_load<float>(((q0 + (unsigned)i) & (RING_PTS - 1)) * 4, LDS_RING_T);
    }
    __device__ __forceinline__ f32x4 c(int i) const {
        return lds_load<f32x4>(((q0 + (unsigned)i) & (RING_PTS - 1)) * 16, LDS_RING_RAW);
    }
};

// one tile's 11 layers; leaves rgb / sigma of the wave's 16 points in lane group 0
__device__ __forceinline__ void run_tile(const Ctx& c, float& sigma, float (&rgb)[3], WFrag& wf) {
    float A[64], B[64];
    run_layer<0>(c, A, A, sigma, rgb, wf);
    run_layer<1>(c, A, B, sigma, rgb, wf);
    run_layer<2>(c, B, A, sigma, rgb, wf);
    run_layer<3>(c, A, B, sigma, rgb, wf);
    run_layer<4>(c, B, A, sigma, rgb, wf);
    run_layer<5>(c, A, B, sigma, rgb, wf);
    run_layer<6>(c, B, A, sigma, rgb, wf);
    run_layer<7>(c, A, B, sigma, rgb, wf);
    run_layer<8>(c, B, A, sigma, rgb, wf);
    run_layer<9>(c, A, B, sigma, rgb, wf);
    run_layer<10>(c, B, A, sigma, rgb, wf);
}

template <bool RAYS, bool COMP>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void nerf_mlp_f32_kernel(MlpArgs a, long long ntiles) {
    static_assert(!COMP || RAYS, "the fused render is a rays-mode kernel");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    (void)smem;
    Ctx c;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.packed), 0,
                                                (int)F32_PACKED_BYTES, 0x00020000);
    c.wave_goff = c.wave * 1024;
    c.lane16 = c.lane * 16;
    c.b_wread = c.lane * 16;
    c.b_wstore = c.wave * 1024 + c.lane * 16;
    c.b_bias = (c.lane >> 4) * 16;
    c.b_posx = LDS_POSX + c.wave * 4096 + c.lane * 16;
    c.b_posd = LDS_POSD + c.wave * 2048 + c.lane * 16;

    {
        const float* bsrc = reinterpret_cast<const float*>(
            reinterpret_cast<const char*>(a.packed) + (long long)F32_WEIGHT_KIB * 1024);
        for (int i = threadIdx.x; i < F32_BIAS_FLOATS; i += WAVES * 64)
            lds_store<float>(i * 4, LDS_BIAS, bsrc[i]);
        Stage<F32_NUM_CHUNKS - 1> st;
        st.load(c);
        st.store(c);
    }
    __syncthreads();
    WFrag wf;                                   // chunk 0's first fragment groups and bias (chunk_step hands them on)
    wf.a[0] = lds_load<f32x4>(c.b_wread, LDS_W0);
    wf.a[1] = lds_load<f32x4>(c.b_wread, LDS_W0 + 1024);
    wf.bias = lds_load<f32x4>(c.b_bias, LDS_BIAS + f32_bias_off(0) * 4);

    if constexpr (!COMP) {
        for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const long long tile_base = tile * TILE_PTS;
            asm volatile("" : "+s"(c.wave_goff));
            stage_inputs<RAYS>(c, a, tile_base, a.P, -1);
            float sigma, rgb[3];
            run_tile(c, sigma, rgb, wf);
            // rows 0..2 (rgb) / row 256 (sigma) are registers 0..2 / 0 of lane group 0
            if (c.lane < 16) {
                const long long p = tile_base + c.wave * 16 + c.lane;
                if (p < a.P) {
                    const f32x4 o = {rgb[0], rgb[1], rgb[2], sigma};
                    *reinterpret_cast<f32x4*>(a.raw + p * 4) = o;
                }
            }
        }
    } else {
        // fused render: this workgroup's contiguous range of rays, tile after tile (mlp_bf16_16.hip COMP)
        const long long B = a.P / a.N;
        const long long r_lo = (long long)blockIdx.x * B / gridDim.x, r_hi = ((long long)blockIdx.x + 1) * B / gridDim.x;
        const long long range_base = r_lo * a.N;
        const int n_pts = (int)((r_hi - r_lo) * a.N);
        const nerf_composite::RayOut out{a.rgb, a.disp, a.alpha, a.acc, a.w, a.pixels};
        int next_ray = 0, n_complete = 0;
        for (int q_tile = 0; q_tile < n_pts; q_tile += TILE_PTS) {
            asm volatile("" : "+s"(c.wave_goff));
            stage_inputs<true>(c, a, range_base + q_tile, range_base + n_pts, q_tile & (RING_PTS - 1));
            float sigma, rgb[3];
            run_tile(c, sigma, rgb, wf);
            if (c.lane < 16) {
                const int local = c.wave * 16 + c.lane;
                if (q_tile + local < n_pts) {
                    const f32x4 o = {rgb[0], rgb[1], rgb[2], sigma};
                    lds_store<f32x4>(((unsigned)(q_tile + local) & (RING_PTS - 1)) * 16, LDS_RING_RAW, o);
                }
            }
            __syncthreads();                                          // every wave's samples of this tile are in the ring
            const int done_q = q_tile + TILE_PTS < n_pts ? q_tile + TILE_PTS : n_pts;
            while ((n_complete + 1) * a.N <= done_q) ++n_complete;
            if (n_complete - next_ray >= WAVES || done_q + TILE_PTS - next_ray * a.N > RING_PTS || done_q == n_pts) {
                for (int ray = next_ray + c.wave; ray < n_complete; ray += WAVES) {
                    const long long gray = r_lo + ray;
                    const float* d = a.rays + gray * 6 + 3;
                    const float dnorm = nerf_composite::unit_dir_norm(d[0], d[1], d[2], true);
                    const RingSamples src{(unsigned)(ray * a.N) & (RING_PTS - 1)};
                    nerf_composite::composite_ray(src, a.N, c.lane, dnorm, gray, out);
                }
                next_ray = n_complete;
                __syncthreads();                                      // the next tile's prologue rewrites ring slots read above
            }
        }
    }
}

}  // namespace

extern "C" int nerf_amd_launch_mlp_f32(const MlpArgs* args, int rays_mode, hipStream_t stream) {
    (void)hipGetLastError();   // drop any stale error: the return value is about THIS launch
    MlpArgs a = *args;
    if (a.P <= 0) return 0;
    const long long ntiles = (a.P + TILE_PTS - 1) / TILE_PTS;
    const int cus = device_cus();
    const long long grid = ntiles < cus ? ntiles : cus;
    const bool comp = a.rgb || a.disp || a.acc || a.alpha || a.w || a.pixels;
    if (comp && (!rays_mode || a.N > FUSED_RENDER_MAX_N || a.P / grid + a.N >= (1ll << 31))) return -2;
    auto kern = comp ? nerf_mlp_f32_kernel<true, true>
                     : (rays_mode ? nerf_mlp_f32_kernel<true, false> : nerf_mlp_f32_kernel<false, false>);
    const int lds = comp ? LDS_TOTAL_COMP : LDS_TOTAL;
    const hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * 64), lds, stream, a, ntiles);
    return (int)hipGetLastError();
}
