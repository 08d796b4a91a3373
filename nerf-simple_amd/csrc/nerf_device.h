// nerf_device.h -- device helpers shared by the gfx950 kernels:
// vector typedefs, the counter RNG, and per-point input assembly
// (reference utils/rendering.py:24-40).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "nerf_layout.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define NERF_FLAG_TS_GIVEN 1u
#define NERF_FLAG_DEVICE_RNG 2u
#define NERF_FLAG_SEED_IN_MEMORY 4u     // with DEVICE_RNG: `u` is the device address of a uint64 added to `seed` at run time
#define NERF_FLAG_STORE_E4M3 8u         // training forward: save the activations in the 8-bit storage form (nerf_layout.h)

// Arguments of the fused sampling + encoding + MLP kernels.
struct MlpArgs {
    const float* pts;     // points mode: [P,6]; rays mode: NULL
    const float* rays;    // rays mode: [B,6]
    const float* u;       // rays mode: jitter or ts [B,N] (NULL with DEVICE_RNG)
    const float* tbins;   // rays mode: [N+1]
    const void* packed;   // packed weight image
    float* raw;           // out [P,4]
    float* ts_out;        // rays mode: out [B,N] (may be NULL)
    void* acts;           // training forward: saved bf16 activations (nerf_layout::acts_total_bytes), else NULL
    long long P;          // points = B*N
    long long ray_id0;    // global id of ray 0 (device RNG counter base)
    unsigned long long seed;
    int N;
    unsigned flags;
    // fused render (sampling + MLP + compositing in one launch, mlp_bf16_16.hip COMP): per-ray outputs,
    // any of them NULL; raw / ts_out are not written in that mode
    float* rgb;           // [B,3]
    float* disp;          // [B]
    float* alpha;         // [B,N]
    float* acc;           // [B]
    float* w;             // [B,N]
    float* pixels;        // [B,4] = [clip(rgb,0,1), disparity]
};

// ---- host: per-device launch facts, looked up once ----------------------------------
// Every launcher used to call hipGetDevice + hipDeviceGetAttribute + hipFuncSetAttribute per launch: a few
// microseconds each on the eager small-batch path.  Both answers are constants of (device) and (kernel, device).
constexpr int NERF_MAX_DEVICES = 64;
inline int device_cus() {
    static int cache[NERF_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NERF_MAX_DEVICES) return 256;
    if (cache[dev] == 0) {
        int cus = 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cache[dev] = cus;                      // racing threads store the same value
    }
    return cache[dev];
}
// hipFuncAttributeMaxDynamicSharedMemorySize for `kernel` on the current device, set once per (kernel, device).
inline hipError_t allow_dynamic_lds(const void* kernel, int bytes) {
    struct Slot { const void* kernel; int dev; };
    static Slot slots[128];
    static std::atomic<int> used{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    const int n = used.load(std::memory_order_acquire);
    for (int i = 0; i < n && i < 128; ++i)
        if (slots[i].kernel == kernel && slots[i].dev == dev) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0) {
        const int i = used.load(std::memory_order_relaxed);
        if (i < 128) {                          // a racing thread may overwrite the slot: the loser just sets the attribute again later
            slots[i] = Slot{kernel, dev};
            used.store(i + 1, std::memory_order_release);
        }
    }
    return e;
}

// ---- chunk barrier of the MLP kernels --------------------------------------------
// Loads, LDS-DMA and stores share vmcnt on gfx9-family parts and retire in issue order, so
// "this wave's DMA pieces have landed" is vmcnt(N) with N = the vector-memory instructions the
// wave issued between its (last) DMA piece of this chunk and the wait (activation / dY / mask stores,
// mask loads).
// Waiting for vmcnt(0) instead -- which __syncthreads() also does through its release fence --
// stalls every chunk on the round trip of the stores just issued.  N is a compile-time count;
// tests/test_library_cpu.py::test_counted_vmcnt_waits checks it against the generated ISA.
constexpr int LOFF_INVALID = 0x40000000;      // lane offset of a point past the end: out of any buffer range
// Which of the 4 feature chunks (16 B = 8 features) of a 32-feature fragment lane group g holds
// once the 8-byte pieces have been traded with v_permlane16_swap (mlp_bf16_16.hip
// epilogue_piece): g = 0,1,2,3 -> chunk 0,2,1,3.
__host__ __device__ constexpr int swapped_chunk(int g) { return (g & 1) * 2 + (g >> 1); }
// byte offset of (that chunk, point `local` of the tile) inside fragment Q = 0's part of a
// point-blocked activation block (nerf_layout.h); fragment Q adds Q * 4 chunks = Q * 16 KiB
__host__ __device__ constexpr int block_lane_offset(int g, int local) { return swapped_chunk(g) * 4096 + local * 16; }

// The epilogue's granule store.  w = {first piece (2 dwords), second piece (2 dwords)} of this lane;
// v_permlane16_swap trades one piece with lane group g ^ 1 (see mlp_bf16_16.hip epilogue_piece) and
// the lane stores 16 contiguous bytes at voffset + soffset of the (layer, tile) block.
//
// Hazard found on MI355X (tools/stress_train.py; it cost NaNs in dY9 in every repetition of the
// backward): a buffer_store_dwordx4 whose soffset is an SGPR, followed IMMEDIATELY by a VALU write
// of its first data register, stored that new value for the last lanes of each 16-lane row.
// hipcc pads this store-data write-after-read case with s_nop only when soffset is not a
// register (its hazard table says a register soffset delays the next instruction enough), so the
// padding is explicit here: the trailing asm keeps the four data registers live past the store
// and supplies the wait states.
// AUX: cache policy bits of the store (0 = default, 2 = non-temporal).  A training step streams 1.3 GB of
// activations and 1.2 GB of dY that nothing reads before they have left every cache, and the dW kernel reads
// each of those bytes once.  Measured on the whole step (4096 x 64, bench.py --mode train, one box): default
// policy everywhere 1.37-1.40 ms; forward stores non-temporal 1.31; + dX-chain stores 1.23; + dW's LDS-DMA loads
// 1.18 ms.  (Timed alone, the dX kernel is 2 % SLOWER with non-temporal stores: the gain is what the
// neighbouring kernels no longer lose to its write-allocated lines.)
template <int AUX = 0>
__device__ __forceinline__ void store_granule(__amdgpu_buffer_rsrc_t rs, int voffset, int soffset, u32x4 w) {
    const auto s0 = __builtin_amdgcn_permlane16_swap(w[0], w[2], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(w[1], w[3], false, false);
    const unsigned o0 = s0[0], o1 = s1[0], o2 = s0[1], o3 = s1[1];
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{o0, o1, o2, o3}, rs, voffset, soffset, AUX);
    asm volatile("s_nop 1" ::"v"(o0), "v"(o1), "v"(o2), "v"(o3) : "memory");
}

// ---- 8-bit storage form (nerf_layout.h): fragments of the wave's 32 points -> e4m3 + the exponent of their group ---------
// Running maximum of the magnitudes that go into a fragment, taken on the fp32 values before they are packed: one
// v_max3_f32 per packed pair (|.| is an operand modifier).  ABS = false is for ReLU outputs: max3(m, v0, v1) with m >= 0
// is the maximum of the rectified values.  (A bf16 rounding may carry the largest value up to the next power of two: the
// stored magnitude is then exactly 256 under the exponent chosen below -- e4m3 holds up to 448.)
template <bool ABS>
__device__ __forceinline__ float f8_absmax(float running, float v0, float v1) {
    float r;
    if constexpr (ABS) asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(running), "v"(v0), "v"(v1));
    else asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(running), "v"(v0), "v"(v1));
    return r;
}
// The wave's maximum of a non-negative per-lane value, as the exponent field of lane 63's result: v_max_f32 with DPP
// operands inside the 16-lane rows, then row_bcast 15 / 31, then one v_readlane.  Written as one asm block: a DPP
// operand read needs two wait states behind the VALU write of that register, which the compiler cannot pad inside
// inline asm (and its own expansion of the builtins costs a copy, a canonicalising maximum and the maximum per step).
__device__ __forceinline__ int f8_wave_max_exponent(float m) {
    int r;
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_readlane_b32 %0, %1, 63"
                 : "=s"(r), "+v"(m));
    return (r >> 23) & 0xff;
}
// two packed bf16 pairs -> four e4m3 bytes, each value divided by `scale` (a power of two) and rounded to nearest even
__device__ __forceinline__ unsigned f8_cvt4(unsigned lo, unsigned hi, float scale) {
    unsigned r;
    asm("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %3\n\tv_cvt_scalef32_pk_fp8_bf16 %0, %2, %3 op_sel:[0,0,1]"
        : "=&v"(r) : "v"(lo), "v"(hi), "s"(scale));
    return r;
}
// w0 / w1 below: a fragment's four packed words of column block 0 / 1 (word j = features 32Q + 16 (j >> 1) + 4 g + 2 (j & 1)
// + {0, 1} of the lane's point, g = lane >> 4).  The wave agrees on the exponent of its largest magnitude (six DPP maxima:
// inside the 16-lane rows, then row_bcast 15 / 31, and one v_readlane), every pair is divided by 2^(exponent - 7) and
// rounded to e4m3 by v_cvt_scalef32_pk_fp8_bf16 (round to nearest even; the largest value lands in [128, 256], nothing
// reaches e4m3's 448), and a 4 x 4 transpose of the lanes' four dwords across the lane groups (two v_permlane32_swap +
// two v_permlane16_swap) leaves lane group g with ONE whole 16-byte granule: groups 0 / 1 = chunks 2Q / 2Q+1 of points
// 0..15 of the wave, groups 2 / 3 the same chunks of points 16..31.  One dwordx4 store per lane: 512 contiguous bytes
// per chunk.
//   rs_data: the (layer, tile) block; voff = f8_lane_offset(lane, wave) or LOFF_INVALID; soff_data = Q * 8192.
//   rs_scale: the tile's 64 exponent bytes of this layer.
__host__ __device__ constexpr int f8_lane_offset(int lane, int wave) {
    return ((lane >> 4) & 1) * 4096 + (wave * 32 + 16 * (lane >> 5) + (lane & 15)) * 16;
}
// One fragment (32 features x the wave's 32 points) under an exponent already agreed on: eight conversions, the 4 x 4
// transpose across the lane groups, one dwordx4 store per lane.
template <int AUX = 0>
__device__ __forceinline__ void store_fragment_f8(__amdgpu_buffer_rsrc_t rs_data, int voff, int soff_data, u32x4 w0, u32x4 w1,
                                                  float scale) {
    const unsigned a0 = f8_cvt4(w0[0], w0[1], scale), b0 = f8_cvt4(w0[2], w0[3], scale);
    unsigned a1 = f8_cvt4(w1[0], w1[1], scale), b1 = f8_cvt4(w1[2], w1[3], scale);
    // v_permlane*_swap reads a VALU result two wait states behind its write; the conversions sit inside inline asm, where
    // the compiler's hazard recogniser does not see them
    asm volatile("s_nop 1" : "+v"(a1), "+v"(b1));
    // lane group g holds piece g of four granules (a0: chunk 2Q of its cb-0 point, b0: chunk 2Q+1, a1 / b1: the cb-1
    // point); after the transpose it holds pieces 0..3 of granule g
    const auto x0 = __builtin_amdgcn_permlane32_swap(a0, a1, false, false);
    const auto x1 = __builtin_amdgcn_permlane32_swap(b0, b1, false, false);
    const auto y0 = __builtin_amdgcn_permlane16_swap(x0[0], x1[0], false, false);
    const auto y1 = __builtin_amdgcn_permlane16_swap(x0[1], x1[1], false, false);
    const unsigned o0 = y0[0], o1 = y0[1], o2 = y1[0], o3 = y1[1];
#if defined(F8_TIMING) && F8_TIMING == 2      // timing-only build (nothing is stored): what do the stores cost?
    asm volatile("" ::"v"(o0), "v"(o1), "v"(o2), "v"(o3));
#else
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{o0, o1, o2, o3}, rs_data, voff, soff_data,
#if defined(F8_TIMING) && F8_TIMING == 3    // timing-only build: cached instead of non-temporal stores
                                           0);
#else
                                           AUX);
#endif
    asm volatile("s_nop 1" ::"v"(o0), "v"(o1), "v"(o2), "v"(o3) : "memory");      // store_granule's hazard
#endif
}
// A GROUP of four consecutive fragments (128 features x the wave's 32 points) under ONE exponent: the cross-lane step
// (fourteen issue slots with its wait states) is paid once per four fragments -- these kernels are bound by vector-
// instruction issue.  The four exponent bytes of the group (equal) go out as one dword by lane 0:
// five vector-memory instructions per group.  w0[k] / w1[k]: fragment 4 (Q / 4) + k of column block 0 / 1; amax: f8_absmax over all
// of them; soff_data: the group's first fragment (Q0 * 8192); scale_byte: wave * 8 + Q0 (a multiple of four).
constexpr int F8_GROUP = 4;
template <int AUX = 0>
__device__ __forceinline__ void store_group_f8(__amdgpu_buffer_rsrc_t rs_data, int voff, int soff_data,
                                               __amdgpu_buffer_rsrc_t rs_scale, int lane, int scale_byte,
                                               const u32x4 (&w0)[F8_GROUP], const u32x4 (&w1)[F8_GROUP], float amax) {
#if defined(F8_TIMING) && F8_TIMING == 1      // timing-only build (wrong exponents): what does the cross-lane chain cost?
    const int e = (__builtin_bit_cast(int, amax) >> 23) & 0xff;
#else
    const int e = f8_wave_max_exponent(amax);                     // biased exponent of the wave's largest magnitude
#endif
    const int sb = e > 8 ? e - 7 : 1;                             // e8m0 byte of the block; >= 1 so the divisor is a normal float
    const float scale = __builtin_bit_cast(float, sb << 23);
#pragma unroll
    for (int k = 0; k < F8_GROUP; ++k) store_fragment_f8<AUX>(rs_data, voff, soff_data + k * 8192, w0[k], w1[k], scale);
#if !(defined(F8_TIMING) && F8_TIMING == 2)
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)sb * 0x01010101u, rs_scale, lane == 0 ? scale_byte : LOFF_INVALID, 0, 0);
#endif
}

template <int N>
__device__ __forceinline__ void chunk_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");            // nothing below is hoisted above the barrier
}

// ---- counter RNG: Philox-4x32-10 (Salmon et al. 2011), keyed by seed,
// counter = global sample id.  One 32-bit word -> u in [0,1) with 24 bits, the
// same granularity as torch.rand for float32.
__device__ __forceinline__ unsigned philox_word(unsigned long long seed, unsigned long long ctr) {
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0x6e657266u, c3 = 0x616d6421u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned long long ctr) {
    return (float)(philox_word(seed, ctr) >> 8) * (1.0f / 16777216.0f);
}

// ---- one IEEE operation, one rounding -------------------------------------------------------------
// hipcc compiles with -ffp-contract=fast-honor-pragmas and its __fmul_rn / __fadd_rn / __fsub_rn are plain `a * b`,
// `a + b`: a product that feeds a sum becomes ONE v_fma / v_fmac whatever the spelling, and __fsqrt_rn is the bare
// v_sqrt_f32 (1 ulp, not correctly rounded).  Where the reference's result has to be reproduced bit for bit -- sample
// positions feed sin(2^9 x): one ulp of y is 1e-4 rad there, and a high-gain network turns that into 5e-5 of alpha --
// the operations are these: the pragma keeps the `contract` flag off the instruction, so nothing fuses with it.
// (Found by the error model of tests/error_model.py: the fp32 render sat 40x further from the float64 value than the
// reference's own fp32 result, the fp32 MLP on given points did not.)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float sqrt_rn(float a) { return sqrtf(a); }       // correctly rounded (v_sqrt + one correction step)

__device__ __forceinline__ float norm3(float x, float y, float z) {
    return sqrt_rn(__fmaf_rn(z, z, __fmaf_rn(y, y, mul_rn(x, x))));
}

// One query point: position (un-normalised direction times t) and the unit
// direction, plus the sample position t (rays mode).
struct PointIn {
    float x, y, z, d1, d2, d3, t;
};

// rays mode (reference utils/rendering.py:24-40).  Every product/sum is
// rounded separately (mul_rn / add_rn above: no fma contraction) exactly like
// the reference's separate torch ops, so ts and locs are bit-identical to the
// CPU path given the same u.
// (ray, sample) of point p = base + local, where (b0, r0) = divmod(base, N) is known (one wave-uniform
// 64-bit division per tile instead of one ~80-instruction division per lane) and local < 2^16:
// r0 + local < N + 2^16 needs only a small quotient, taken from a float reciprocal and corrected.
struct RaySample { long long b; int i; };
__device__ __forceinline__ RaySample split_point(long long b0, int r0, int local, int N) {
    const unsigned tt = (unsigned)(r0 + local);
    unsigned q = (unsigned)((float)tt * __frcp_rn((float)N));
    if ((long long)q * N > (long long)tt) --q;                 // the float estimate is off by at most one
    if ((long long)(q + 1) * N <= (long long)tt) ++q;
    return RaySample{b0 + q, (int)(tt - q * (unsigned)N)};
}
__device__ __forceinline__ RaySample split_point(long long p, int N) {
    const long long b = p / N;
    return RaySample{b, (int)(p - b * N)};
}

// The seed of the counter RNG.  A launch captured into a hipGraph is replayed with its arguments frozen; with
// NERF_FLAG_SEED_IN_MEMORY the `u` argument is not jitter but the device address of a 64-bit word that is added to the
// seed when the kernel RUNS, so every replay of a training step draws fresh jitter (one scalar load per use).
__device__ __forceinline__ unsigned long long effective_seed(const MlpArgs& a) {
    if (!(a.flags & NERF_FLAG_SEED_IN_MEMORY)) return a.seed;
    const unsigned long long v = *reinterpret_cast<const unsigned long long*>(a.u);
    // wave-uniform by construction: keep it in scalar registers whatever load the compiler picked
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return a.seed + (((unsigned long long)hi << 32) | lo);
}

// the counter RNG's draw for (ray b, sample i) of this launch (FLAG_DEVICE_RNG)
__device__ __forceinline__ float device_rng_uniform(const MlpArgs& a, RaySample rs) {
    // The seed passes through an empty asm so that the ten rounds' key schedule (k += const, wave-uniform)
    // is recomputed here -- 20 scalar adds -- instead of being hoisted out of the persistent tile loop as
    // 20 live SGPRs, which the fused render kernel could only keep by spilling them to VGPR lanes
    // (v_writelane / v_readlane + hazard nops inside every Philox round).
    unsigned long long seed = effective_seed(a);
    asm volatile("" : "+s"(seed));
    return philox_uniform(seed, (unsigned long long)((a.ray_id0 + rs.b) * a.N + rs.i));
}

// u_pre: the point's device-RNG draw when the caller already has it (have_u), see mlp_bf16_16.hip stage_inputs
// DIR = false: the unit direction is left out (d1..d3 = 0) for callers that encode it once per ray instead
template <bool DIR = true>
__device__ __forceinline__ PointIn fetch_point_rays(const MlpArgs& a, long long p, RaySample rs, float u_pre = 0.f,
                                                    bool have_u = false) {
    PointIn r;
    const long long b = rs.b;
    const int i = rs.i;
    const float* ray = a.rays + b * 6;
    const float ox = ray[0], oy = ray[1], oz = ray[2];
    const float dx = ray[3], dy = ray[4], dz = ray[5];
    float t;
    // jitter / sample positions are read once: non-temporal loads keep 4 B per sample out of the caches
    if (a.flags & NERF_FLAG_TS_GIVEN) {
        t = __builtin_nontemporal_load(a.u + p);
    } else {
        float u;
        if (a.flags & NERF_FLAG_DEVICE_RNG)
            u = have_u ? u_pre : philox_uniform(effective_seed(a), (unsigned long long)((a.ray_id0 + b) * a.N + i));
        else
            u = __builtin_nontemporal_load(a.u + p);
        const float bin_diff = sub_rn(a.tbins[1], a.tbins[0]);
        t = add_rn(mul_rn(bin_diff, u), a.tbins[i]);
    }
    r.t = t;
    r.x = add_rn(ox, mul_rn(dx, t));
    r.y = add_rn(oy, mul_rn(dy, t));
    r.z = add_rn(oz, mul_rn(dz, t));
    if constexpr (DIR) {
        // torch.norm over 3 elements on CPU == sqrt(fma(z,z,fma(y,y,x*x))) bit for bit
        const float nrm = norm3(dx, dy, dz);
        r.d1 = __fdiv_rn(dx, nrm);
        r.d2 = __fdiv_rn(dy, nrm);
        r.d3 = __fdiv_rn(dz, nrm);
    } else {
        r.d1 = r.d2 = r.d3 = 0.f;
    }
    return r;
}

__device__ __forceinline__ PointIn fetch_point_rays(const MlpArgs& a, long long p) {
    return fetch_point_rays(a, p, split_point(p, a.N));
}

__device__ __forceinline__ PointIn fetch_point_pts(const MlpArgs& a, long long p) {
    PointIn r;
    const float* v = a.pts + p * 6;
    r.x = v[0]; r.y = v[1]; r.z = v[2];
    r.d1 = v[3]; r.d2 = v[4]; r.d3 = v[5];
    r.t = 0.f;
    return r;
}

// x / (2 pi) as an unevaluated sum hi + lo (|lo| <= ulp(hi)/2): scaling by 2^k
// is exact and fract(hi * 2^k) is exact, so the phase of sin(2^k x) keeps
// ~1e-7 rad accuracy even at |2^k x| ~ 2300 rad.
struct TwoF { float hi, lo; };
__device__ __forceinline__ TwoF to_revolutions(float x) {
    const float C_HI = 0.15915494f;          // fl32(1/(2 pi))
    const float C_LO = 6.4206382e-09f;       // 1/(2 pi) - C_HI
    TwoF q;
    q.hi = mul_rn(x, C_HI);
    const float err = __fmaf_rn(x, C_HI, -q.hi);
    q.lo = __fmaf_rn(x, C_LO, err);
    return q;
}
// sin / cos of (2 pi * 2^k * q) via the hardware v_sin/v_cos (argument in
// revolutions).  Accuracy ~1e-6 abs: used by the bf16 path only.
__device__ __forceinline__ void sincos_rev_fast(TwoF q, float scale, float& s, float& c) {
    const float f = __builtin_amdgcn_fractf(q.hi * scale) + q.lo * scale;
    s = __builtin_amdgcn_sinf(f);
    c = __builtin_amdgcn_cosf(f);
}
