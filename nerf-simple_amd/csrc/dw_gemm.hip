// dw_gemm.hip -- parameter gradients of the dense layers (training step,
// reference train.py:51-54: loss.backward() as far as the 24 parameter tensors).
//
//   dW_l = dY_l^T @ X_l      (sum over all P = B*N query points)
//   db_l = sum_p dY_l[p, :]  (column sums of the A operand's LDS slabs, same kernel)
//
// dY_l (from nerf_amd_mlp_backward) and X_l (the activations saved by
// nerf_amd_mlp_forward_train, point-blocked: nerf_layout.h; plus the row-major encoder
// outputs) have the reduction index (the point) as their slow dimension in both MFMA
// operands: a "TN" GEMM with M, N <= 256 and K = P ~ 10^5..10^6.
// The vendor library runs this shape on 16 workgroups; here:
//   * ONE launch covers all 14 products; each gets a share of the ~256
//     workgroups proportional to the bytes it streams (split-K over the points);
//   * a workgroup (8 waves) owns the full 256x256 output of its product in
//     registers and walks its K slice in slabs of 32 points, moved HBM -> LDS by
//     LDS-DMA (no staging registers) into a 4-slot ring: three slabs (96 KiB per
//     CU) stay in flight behind counted vmcnt waits and one raw s_barrier per slab;
//   * the LDS images (one form per operand layout, below) are unpadded with an XOR
//     swizzle of the 16-byte granules (applied to the DMA's per-lane source and to the reads);
//   * fragments come out of LDS through ds_read_b64_tr_b16, the hardware
//     transposing read: 4 points x 16 features in, 4 consecutive k per lane out,
//     so no transpose pass exists anywhere;
//   * partial tiles are added into ONE flat fp32 gradient vector (state_dict
//     order, the all-reduce bucket) with float atomics, one 32x32 accumulator
//     register = two 128-B row segments per wave-instruction.
// HBM-bound by design: every dY / X byte is read once per product (~11.5 KB per
// point in all); 128 FLOP per byte.
#include "nerf_device.h"

using namespace nerf_layout;

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAXD = 16;
struct GemmDesc {
    const __bf16* A;   // [P, lda] row-major, or (lda == 0) one point-blocked activation layer; columns 0..M-1 -> output ROWS
    const __bf16* B;   // [P, ldb] row-major, or (ldb == 0) point-blocked;                          columns 0..N-1 -> output COLUMNS
    float* C;          // destination of output element (r0, 0)
    int lda, ldb, ldc;
    int M, N;          // operand widths actually read (multiples of 32)
    int r0, Mv, Nv;    // rows [r0, r0+Mv) x cols [0, Nv) are stored
    int wg0, wgs;      // workgroups [wg0, wg0+wgs) split the K range
    float* bias;       // non-NULL: also add the column sums of A[:, 0..M) (= db of that layer) here
};
struct GemmTable {
    GemmDesc d[MAXD];
    int n;
    long long P;
};

constexpr int SLAB = 32;                       // points per LDS slab (2 k-steps of 16)
static_assert(SLAB == 32, "the slab loop is written for two k-steps");
static_assert(ACT_TILE_PTS % SLAB == 0 && ACT_TILE_PTS * 16 == 4096, "a slab never straddles activation blocks");
constexpr int ROWB = 512;                      // LDS row = 256 bf16 features, unpadded (LDS-DMA is lane-linear)
constexpr int OPB = SLAB * ROWB;               // one operand slab = 16 KiB
constexpr int KSTEP = 16 * ROWB;               // second k-step (points 16..31) of either LDS image form
constexpr int SLOTB = 2 * OPB;                 // A + B
constexpr int RING = 4;                        // slabs resident in LDS: one computing, up to three landing
constexpr int LDS_BYTES = RING * SLOTB;        // 128 KiB
constexpr int DMA_PER_SLAB = 2 * SLAB / 2 / 8; // 1 KiB wave-instructions per wave per slab (8 waves) = 4

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lds_void;

// LDS image of one operand slab (32 points x 32 chunks of 16 B = 16 KiB), two forms, both filled
// by lane-linear LDS-DMA with the swizzle applied to each lane's SOURCE granule and again to the
// read address (cdna_hip_programming.md section 5.4 rule 21), both conflict-free for
// ds_read_b64_tr_b16 (the 32 lanes of a half-wave touch 32 distinct 8-byte words of a 256-byte
// bank row) and both with the second k-step 8 KiB after the first:
//   row-major operand (posx / posd / d_raw rows):  granule (point r, chunk c) at
//       r * 512 + (c ^ ((r & 3) << 2)) * 16                 -- a DMA instruction = 2 rows;
//   point-blocked operand (activations, dY):        granule (point r = 16 ks + r', chunk c) at
//       ks * 8192 + c * 256 + (r' ^ ((c & 3) << 2)) * 16    -- a DMA instruction = 4 chunks x 16
//       points, i.e. four 256-byte contiguous runs of the block in HBM.
// The transposing reads are issued through inline asm: hipcc treats the ds_read_tr builtin as a
// possible alias of every LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of it, which
// would drain the three-slab prefetch each slab.  The asm reads are invisible to the compiler's
// counters, so their lgkmcnt wait is explicit too (frags_landed ties the registers to the wait).
struct Frag { bf16x4 lo, hi; };                    // k = 8h + 0..3 and 8h + 4..7 of this lane's column
template <int KOFF>
__device__ __forceinline__ void read_frag_tr(Frag& f, unsigned addr_lo, unsigned addr_hi) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"
                 : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr_lo), "v"(addr_hi), "n"(KOFF));
}
__device__ __forceinline__ void frags_landed(Frag (&a)[4], Frag (&b)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0].lo), "+v"(a[0].hi), "+v"(a[1].lo), "+v"(a[1].hi), "+v"(a[2].lo), "+v"(a[2].hi),
                   "+v"(a[3].lo), "+v"(a[3].hi), "+v"(b[0].lo), "+v"(b[0].hi), "+v"(b[1].lo), "+v"(b[1].hi));
}
__device__ __forceinline__ bf16x8 whole(const Frag& f) {
    return bf16x8{f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
}

__global__ __launch_bounds__(512, 2) void dw_gemm_kernel(GemmTable tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // which product, which K slice
    int di = 0;
    while (di + 1 < tab.n && (int)blockIdx.x >= tab.d[di + 1].wg0) ++di;
    const GemmDesc d = tab.d[di];
    const int slice = blockIdx.x - d.wg0;
    const long long nslab = (tab.P + SLAB - 1) / SLAB;
    const long long s_begin = nslab * slice / d.wgs, s_end = nslab * (slice + 1) / d.wgs;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves over the 256 x 256 tile
    const int m0 = 128 * wm, n0 = 64 * wn;            // this wave: 4 x 2 tiles of 32 x 32

    // ---- LDS-DMA geometry: wave w issues pieces 4w .. 4w+3 of a slab (pieces 0..15: A, 16..31: B);
    // a piece is 1 KiB of the operand's LDS image, lane l at byte 16 l of it
    const bool pieceA = wave < 4;
    const int pld = pieceA ? d.lda : d.ldb;           // 0: point-blocked
    const bool pblk = pld == 0;
    const __bf16* pbase = pieceA ? d.A : d.B;
    const int pwidth = pieceA ? d.M : d.N;
    const long long row_begin = s_begin * SLAB;
    const long long row_end = s_end * SLAB < tab.P ? s_end * SLAB : tab.P;
    const long long slice_rows = row_end > row_begin ? row_end - row_begin : 0;
    // Both forms: descriptor rebased to this workgroup's K slice (below 1 GiB: launcher), every
    // offset in the per-lane VGPR offset, which is what the range check sees for certain.
    // Row-major: the range ends at min(P, slice end), so rows past the operand return zeros.
    // Point-blocked: the range covers the slice's blocks; lanes of points >= P are pointed outside
    // by hand (their granules lie between valid ones).
    const long long slice_blocks = (row_end + ACT_TILE_PTS - 1) / ACT_TILE_PTS - row_begin / ACT_TILE_PTS;
    const __amdgpu_buffer_rsrc_t rs = pblk
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(pbase))
                                                + (row_begin / ACT_TILE_PTS) * ACT_BLOCK_BYTES,
                                            0, (int)(slice_rows > 0 ? slice_blocks * ACT_BLOCK_BYTES : 0), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(pbase + row_begin * pld), 0,
                                            (int)(slice_rows * pld * 2), 0x00020000);
    constexpr unsigned OUTSIDE = 0xffffffffu;         // >= num_records: the lane loads zeros
    unsigned voff[DMA_PER_SLAB];                      // per-lane source offset inside a slab
    unsigned ldst[DMA_PER_SLAB];                      // wave-uniform destination inside a ring slot
    int vrow[DMA_PER_SLAB];                           // point (row of the slab) the lane fetches
#pragma unroll
    for (int k = 0; k < DMA_PER_SLAB; ++k) {
        const int piece = wave * DMA_PER_SLAB + k, pi = piece & 15;
        ldst[k] = (piece >> 4) * OPB + pi * 1024;
        if (pblk) {
            const int c = 4 * (pi & 7) + (lane >> 4);                              // chunk
            vrow[k] = 16 * (pi >> 3) + ((lane & 15) ^ ((lane >> 4) << 2));         // (c & 3) == lane >> 4
            voff[k] = c * 8 < pwidth ? (unsigned)(c * (ACT_TILE_PTS * 16) + vrow[k] * 16) : OUTSIDE;
        } else {
            vrow[k] = 2 * pi + (lane >> 5);
            const int cg = (lane & 31) ^ ((vrow[k] & 3) << 2);                     // the chunk this LDS position holds
            voff[k] = cg * 8 < pwidth ? (unsigned)(vrow[k] * pld * 2 + cg * 16) : OUTSIDE;
        }
    }
    auto issue_slab = [&](int it, int slot) {         // it = slab index inside the slice
        const long long p0 = row_begin + (long long)it * SLAB;
        unsigned vadd;
        int rows_left = SLAB;
        if (pblk) {
            vadd = (unsigned)((p0 / ACT_TILE_PTS - row_begin / ACT_TILE_PTS) * ACT_BLOCK_BYTES + (p0 % ACT_TILE_PTS) * 16);
            rows_left = tab.P - p0 < SLAB ? (int)(tab.P - p0) : SLAB;
        } else {
            vadd = (unsigned)it * (unsigned)(SLAB * pld * 2);
        }
#pragma unroll
        for (int k = 0; k < DMA_PER_SLAB; ++k) {
            lds_void* dst = reinterpret_cast<lds_void*>(reinterpret_cast<lds_char*>(0) + slot * SLOTB + ldst[k]);
            const unsigned off = (voff[k] == OUTSIDE || vrow[k] >= rows_left) ? OUTSIDE : voff[k] + vadd;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, off, 0, 0, 2);      // aux 2 = non-temporal: every operand byte is read once
        }
    };

    // ---- fragment addresses (first / second transposing read), per operand form.  The lane reads
    // points 8h + q (+4) of the k-step for features 32 t + 16 (group & 1) + 4 pp .. +3 of tile t,
    // i.e. chunk 4 t + 2 (group & 1) + (pp >> 1), bytes 8 (pp & 1) .. +7 of the granule.
    const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = lane >> 5;
    const int cl = 2 * (grp & 1) + (pp >> 1);          // chunk inside the tile's 4
    auto frag_addr = [&](bool blocked, int t, int second) -> unsigned {
        const int r = 8 * h + q + 4 * second;
        if (blocked) return (4 * t + cl) * 256 + ((r ^ (cl << 2)) * 16) + 8 * (pp & 1);
        return r * ROWB + ((((4 * t + cl) ^ ((r & 3) << 2))) * 16) + 8 * (pp & 1);
    };
    unsigned aoff[4][2], boff[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) aoff[i][e] = frag_addr(d.lda == 0, m0 / 32 + i, e);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) boff[j][e] = OPB + frag_addr(d.ldb == 0, n0 / 32 + j, e);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // waves whose tile rows / columns lie outside the product skip the arithmetic
    const bool active = m0 < d.M && n0 < d.N;
    // db = column sums of dY (the A operand, point-blocked): thread -> chunk tid >> 4, points (tid & 15) + 16 ks,
    // so a quarter-wave reads 256 contiguous bytes
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int brow = tid & 15, bchunk = tid >> 4;
    const unsigned baddr = bchunk * 256 + ((brow ^ ((bchunk & 3) << 2)) * 16);

    // prologue: up to RING-1 slabs in flight; slab 0 published, its first fragments requested
#pragma unroll
    for (int k = 0; k < RING - 1; ++k)
        if (s_begin + k < s_end) issue_slab(k, k);
    // The slab loop is software-pipelined across slabs.  A slab's barrier -- "everyone's pieces of slab s+1 have landed,
    // everyone is past slab s-1" -- sits in the MIDDLE of slab s, behind the eight MFMAs of its first k-step, and the
    // first fragments of slab s+1 are requested right behind the second k-step's fragments have landed: the barrier
    // wait and the LDS round trip of every slab's first reads are covered by matrix work of the same wave instead of
    // standing at the head of each slab (the early-barrier idea of mlp_bf16_16.hip).  Fragment registers: a0/b0 are
    // free once the first k-step's MFMAs are issued, so the next slab's first fragments reuse them.
    Frag a0[4], b0[2], a1[4], b1[2];
    auto read_k0 = [&](unsigned base) {
#pragma unroll
        for (int i = 0; i < 4; ++i) read_frag_tr<0>(a0[i], base + aoff[i][0], base + aoff[i][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) read_frag_tr<0>(b0[j], base + boff[j][0], base + boff[j][1]);
    };
    if (s_begin < s_end) {
        const long long n = s_end - s_begin;
        if (n >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // everyone's pieces of slab 0 landed
        if (active) read_k0(0);
    }
    for (long long s = s_begin; s < s_end; ++s) {
        const int it = (int)(s - s_begin), slot = it & (RING - 1);
        const long long left = s_end - s;
        const unsigned base = slot * SLOTB;
        if (active) {
            frags_landed(a0, b0);                  // slab s, first k-step (requested half a slab ago)
#pragma unroll
            for (int i = 0; i < 4; ++i) read_frag_tr<KSTEP>(a1[i], base + aoff[i][0], base + aoff[i][1]);
#pragma unroll
            for (int j = 0; j < 2; ++j) read_frag_tr<KSTEP>(b1[j], base + boff[j][0], base + boff[j][1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whole(a0[i]), whole(b0[j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (left >= 2) {
            // this wave's pieces of slab s+1 have landed once at most the DMAs of slab s+2 remain (s+3 is not issued yet)
            if (left >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // slab s+1 published; everyone is done with slab s-1 (and its column sums)
            if (s + RING - 1 < s_end) issue_slab(it + RING - 1, (it + RING - 1) & (RING - 1));    // into the slot of slab s-1
        }
        if (active) {
            frags_landed(a1, b1);
            if (left >= 2) read_k0(((it + 1) & (RING - 1)) * SLOTB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whole(a1[i]), whole(b1[j]), acc[i][j], 0, 0, 0);
        }
        if (d.bias && bchunk * 8 < d.M) {
#pragma unroll
            for (int ks = 0; ks < SLAB / 16; ++ks) {
                const bf16x8 v = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(
                    reinterpret_cast<lds_char*>(0) + base + ks * KSTEP + baddr);
#pragma unroll
                for (int k = 0; k < 8; ++k) bsum[k] += (float)v[k];
            }
        }
    }
    __syncthreads();
    if (d.bias && bchunk * 8 < d.M && s_begin < s_end) {
        // 16 threads (brow) hold partial sums of the same 8 columns: combine through LDS, then atomics
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < 8; ++k) red[(brow * 32 + bchunk) * 8 + k] = bsum[k];
    }
    __syncthreads();
    if (d.bias && s_begin < s_end && tid < d.M) {
        const int ch = tid >> 3, k = tid & 7;
        float v = 0.f;
        for (int r = 0; r < 16; ++r) v += reinterpret_cast<float*>(smem)[(r * 32 + ch) * 8 + k];
        atomicAdd(d.bias + tid, v);
    }
    // split-K combine: float atomics into the flat gradient vector
    if (active && s_begin < s_end) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h - d.r0;
                    const int col = n0 + 32 * j + (lane & 31);
                    if (row >= 0 && row < d.Mv && col < d.Nv)
                        atomicAdd(d.C + (long long)row * d.ldc + col, acc[i][j][r]);
                }
    }
}

// d_raw [P,4] fp32 -> dsr [P,32] bf16 (cols 0..2 = drgb, col 3 = dsigma, rest 0) for the two
// head products, and the head bias gradients (sum drgb -> color_fc.2.bias, sum dsigma ->
// sigma_fc.0.bias) straight from the fp32 values.
__global__ __launch_bounds__(256) void pack_draw_kernel(const float* __restrict__ d_raw, __bf16* __restrict__ dsr,
                                                        long long P, float* __restrict__ g_rgb_b,
                                                        float* __restrict__ g_sig_b) {
    __shared__ float red[4][4];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const __bf16 z = (__bf16)0.f;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(d_raw + p * 4);
        s += d;
        bf16x8* o = reinterpret_cast<bf16x8*>(dsr + p * 32);
        o[0] = bf16x8{(__bf16)d[0], (__bf16)d[1], (__bf16)d[2], (__bf16)d[3], z, z, z, z};
        o[1] = bf16x8{z, z, z, z, z, z, z, z};
        o[2] = o[1];
        o[3] = o[1];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = s[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(threadIdx.x < 3 ? g_rgb_b + threadIdx.x : g_sig_b, v);
    }
}

// Zero fill as a KERNEL.  hipMemsetAsync here becomes a memset node when the step is captured into a
// hipGraph, and on ROCm 7.2 the float atomics of the kernels behind that node (executed at the memory
// side) were seen to land on the buffer's OLD contents: gradients of 1e22..1e33 from the fourth
// GraphedTrainStep of a process on, whenever the allocation had a previous tenant (eager launches and
// fresh allocations were fine).  A kernel node orders like every other kernel -> kernel edge.
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long long n) {
    const long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) *reinterpret_cast<f32x4*>(p + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    else for (long long k = i; k < n; ++k) p[k] = 0.f;
}

}  // namespace

// grads: flat fp32 [595844] in state_dict order (zeroed here); scratch: P*64 bytes (dsr).
// begin: zero the gradient vector, pack d_raw for the two head products, head bias gradients.
// finish: the 14 split-K products + the other bias gradients.  (Two entry points so a captured step can
// run `begin` on a side branch beside the dX chain, which needs d_raw as well.)
extern "C" int nerf_amd_launch_param_gradients_begin(const float* d_raw, void* scratch, float* grads, long long P,
                                                     hipStream_t stream) {
    (void)hipGetLastError();
    static_assert(PARAM_COUNT % 4 == 0, "16-byte zero fill");
    hipLaunchKernelGGL(zero_f32_kernel, dim3((PARAM_COUNT / 4 + 255) / 256), dim3(256), 0, stream, grads,
                       (long long)PARAM_COUNT);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (P <= 0) return 0;
    hipLaunchKernelGGL(pack_draw_kernel, dim3(512), dim3(256), 0, stream, d_raw, reinterpret_cast<__bf16*>(scratch), P,
                       grads + OFF_C1_B, grads + OFF_SIG_B);
    return (int)hipGetLastError();
}

// bucket 0: all 14 products in one launch.  Buckets 1 and 2 split them at the boundary of the flat gradient
// vector that data-parallel training reduces in two pieces (nerf_layout.h GRAD_BUCKET_SPLIT): bucket 1 = the LATE
// layers (skip_conn_layer ... color_fc.2, the tail of the vector), bucket 2 = layers_0.* (its head), so the
// all-reduce of bucket 1 runs while bucket 2 is still being computed.  Either launch fills the chip.
extern "C" int nerf_amd_launch_param_gradients_finish(const void* acts_v, const void* dys_v, const void* posx64_v,
                                                      const void* posd32_v, const void* scratch, float* grads,
                                                      long long P, int bucket, hipStream_t stream) {
    (void)hipGetLastError();
    hipError_t e = hipSuccess;
    if (P <= 0) return 0;
    const __bf16* acts = reinterpret_cast<const __bf16*>(acts_v);
    const __bf16* dys = reinterpret_cast<const __bf16*>(dys_v);
    const __bf16* posx = reinterpret_cast<const __bf16*>(posx64_v);
    const __bf16* posd = reinterpret_cast<const __bf16*>(posd32_v);
    const __bf16* dsr = reinterpret_cast<const __bf16*>(scratch);
    // point-blocked activation layers (nerf_layout.h): layer L at L * act_layer_stride(P) bytes; leading dimension 0
    auto act = [&](int L) { return acts + act_offset_bytes(L, P) / 2; };
    auto dy = [&](int L) { return dys + act_offset_bytes(L, P) / 2; };
    constexpr int BLK = 0;

    GemmTable t{};
    t.P = P;
    int n = 0;
    auto add = [&](const __bf16* A, int lda, int M, const __bf16* B, int ldb, int N, int coff, int ldc, int r0,
                   int Mv, int Nv, int boff = -1) {
        GemmDesc& g = t.d[n++];
        g.A = A; g.lda = lda; g.M = M; g.B = B; g.ldb = ldb; g.N = N;
        g.C = grads + coff; g.ldc = ldc; g.r0 = r0; g.Mv = Mv; g.Nv = Nv;
        g.bias = boff >= 0 ? grads + boff : nullptr;      // db of the layer whose dY is this product's A (point-blocked A only)
    };
    const int LW = 256 * 256 + 256;
    add(dy(0), BLK, 256, posx, 64, 64, OFF_L0_W, 63, 0, 256, 63, OFF_L0_B);               // layers_0.0
    for (int l = 1; l <= 4; ++l)                                                          // layers_0.{2,4,6,8}
        add(dy(l), BLK, 256, act(l - 1), BLK, 256, OFF_L1_W + (l - 1) * LW, 256, 0, 256, 256,
            OFF_L1_W + (l - 1) * LW + 65536);
    add(dy(5), BLK, 256, act(4), BLK, 256, OFF_SKIP_W, 319, 0, 256, 256, OFF_SKIP_B);     // skip [h ; x]: h part
    add(dy(5), BLK, 256, posx, 64, 64, OFF_SKIP_W + 256, 319, 0, 256, 63);                //               x part
    add(dy(6), BLK, 256, act(5), BLK, 256, OFF_L6_W, 256, 0, 256, 256, OFF_L6_W + 65536); // layers_1.0
    add(dy(7), BLK, 256, act(6), BLK, 256, OFF_L6_W + LW, 256, 0, 256, 256, OFF_L6_W + LW + 65536);  // layers_1.2
    add(dsr, 32, 32, act(7), BLK, 256, OFF_SIG_W, 256, 3, 1, 256);                        // sigma_fc.0 (row 3 of dsr)
    add(dy(8), BLK, 256, act(7), BLK, 256, OFF_L2_W, 256, 0, 256, 256, OFF_L2_B);         // layers_2
    add(dy(9), BLK, 128, act(8), BLK, 256, OFF_C0_W, 283, 0, 128, 256, OFF_C0_B);         // color_fc.0 [h ; d]: h part
    add(dy(9), BLK, 128, posd, 32, 32, OFF_C0_W + 256, 283, 0, 128, 27);                  //                      d part
    add(dsr, 32, 32, act(9), BLK, 128, OFF_C1_W, 128, 0, 3, 128);                         // color_fc.2 (rows 0..2)
    if (bucket != 0) {
        // keep the products whose destination lies in this bucket's part of the flat vector
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const bool head = t.d[i].C - grads < GRAD_BUCKET_SPLIT;
            if (head == (bucket == 2)) t.d[m++] = t.d[i];
        }
        n = m;
    }
    t.n = n;
    // workgroups per product, one per CU in total.  A slab costs a workgroup the bytes it streams,
    // (M + N) * 2 per point, plus a fixed part (barrier, DMA issue, fragment reads) that measures
    // larger than the byte part: profiles/r01e_dw_sweep.jsonl -- weights (M + N) + c with c = 0 /
    // 256 / 1024 / 8192 give 0.79 / 0.63 / 0.62 / 0.61 ms at 262144 points.  (Sizing by flops
    // left the thin products -- 32 x 256 reads as much of X as 256 x 256 -- as a 2 ms tail.)
    constexpr double slab_cost = 1024.0;
    const int cus = device_cus();
    double total = 0;
    for (int i = 0; i < n; ++i) total += (double)(t.d[i].M + t.d[i].N) + slab_cost;
    const long long nslab = (P + SLAB - 1) / SLAB;
    // exactly `cus` workgroups in total when the slices allow it (the 128 KiB ring leaves one
    // workgroup per CU, so a 257th would run as a second wave and double the kernel's time):
    // floor of each share, then the largest remainders get the leftover workgroups
    const long long need = (P * 512 >> 30) + 1;           // a slice's operand stays below 1 GiB (32-bit buffer range)
    long long w[16];
    double rem[16];
    long long used = 0;
    for (int i = 0; i < n; ++i) {
        const double share = ((double)(t.d[i].M + t.d[i].N) + slab_cost) / total * cus;
        w[i] = (long long)share;
        rem[i] = share - (double)w[i];
        if (w[i] < need) { w[i] = need; rem[i] = 0; }
        if (w[i] >= nslab) { w[i] = nslab; rem[i] = -1; }
        used += w[i];
    }
    while (used < cus) {
        int best = -1;
        for (int i = 0; i < n; ++i)
            if (rem[i] >= 0 && w[i] < nslab && (best < 0 || rem[i] > rem[best])) best = i;
        if (best < 0) break;
        ++w[best]; rem[best] = 0; ++used;      // a second round goes by product order
    }
    int wg = 0;
    for (int i = 0; i < n; ++i) {
        t.d[i].wg0 = wg;
        t.d[i].wgs = (int)w[i];
        wg += (int)w[i];
    }
    e = allow_dynamic_lds(reinterpret_cast<const void*>(dw_gemm_kernel), LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dw_gemm_kernel, dim3(wg), dim3(512), LDS_BYTES, stream, t);

    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_param_gradients(const float* d_raw, const void* acts_v, const void* dys_v,
                                               const void* posx64_v, const void* posd32_v, void* scratch,
                                               float* grads, long long P, hipStream_t stream) {
    const int rc = nerf_amd_launch_param_gradients_begin(d_raw, scratch, grads, P, stream);
    if (rc) return rc;
    return nerf_amd_launch_param_gradients_finish(acts_v, dys_v, posx64_v, posd32_v, scratch, grads, P, 0, stream);
}
