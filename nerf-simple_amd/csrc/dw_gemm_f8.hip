// dw_gemm_f8.hip -- the parameter gradients of dw_gemm.hip from operands in the 8-bit storage form
// (nerf_layout.h: e4m3 + one power-of-two exponent per 32 features x 32 points).
//
//   dW_l = dY_l^T @ X_l,  db_l = sum_p dY_l[p, :]        (reference train.py:51-54: loss.backward())
//
// Same decomposition as dw_gemm.hip -- one launch, every product a share of the ~256 workgroups, a workgroup owns the
// whole 256 x 256 output in registers and walks its slice of the points through a 4-slot LDS ring filled by LDS-DMA --
// with the matrix work on v_mfma_scale_f32_32x32x64_f8f6f4, whose block scales are exactly the stored exponents:
//   * a slab is 64 points = two 32-point exponent blocks = ONE instruction per 32 x 32 output tile.  Its K layout
//     (measured, tools/micro/f8_probe.hip): a lane's bytes 0..15 belong to K block 0, bytes 16..31 to block 1; lanes r
//     and r + 32 hold the two halves of row r; the scale byte of block 0 is read from lane r, that of block 1 from lane
//     r + 32.  So lane (r, h) carries points 16h .. 16h+15 of exponent block 0 and of exponent block 1, and its scale
//     register the exponent of block h;
//   * every operand is point-blocked ([tile][chunk of 16 features][256 points][16 B]): a DMA instruction moves one chunk
//     of the slab, 1 KiB contiguous in HBM, to 1 KiB of LDS; chunks sit 1152 B apart so the two chunks a half-wave
//     reads (features 0..15 / 16..31 of a tile) fall on different halves of the 256-byte bank row;
//   * fragments come out through ds_read_b64_tr_b8 (8 points x 16 features in, 8 consecutive points of one feature per
//     lane out): four reads fill a lane's 32 bytes;
//   * the exponents of the slab's two blocks (8 bytes each per operand) come by scalar loads;
//   * db = A^T . 1: one more MFMA per wave and slab against a fragment of ones.
// HBM-bound by design at half the bytes of the bf16 form: (M + N) bytes per point and product.
#include "nerf_device.h"

using namespace nerf_layout;

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int MAXD = 16;
struct F8Operand {
    const char* data;        // [tile][chunk][256][16]
    const unsigned char* scales;   // [32-point block][8]
    int block_bytes;         // bytes per tile (chunks * 4096)
    int width;               // features present (multiple of 16); chunks beyond read as zeros
};
struct GemmDescF8 {
    F8Operand A, B;          // A's features -> output rows, B's -> output columns
    float* C;                // destination of output element (r0, 0)
    int ldc;
    int r0, Mv, Nv;          // rows [r0, r0 + Mv) x cols [0, Nv) are stored
    int wg0, wgs;            // workgroups [wg0, wg0 + wgs) split the K range
    float* bias;             // non-NULL: also add the column sums of A here
};
struct GemmTableF8 {
    GemmDescF8 d[MAXD];
    int n;
    long long P;
};

constexpr int SLAB = 64;                        // points per slab: two exponent blocks, one MFMA per output tile
constexpr int CHS = 1152;                       // LDS stride of a chunk image (1 KiB + 128 B)
constexpr int OPB = 16 * CHS;                   // one operand slab in LDS
constexpr int SLOTB = 2 * OPB;
constexpr int RING = 4;
constexpr int LDS_BYTES = RING * SLOTB;         // 144 KiB
constexpr int DMA_PER_SLAB = 4;                 // wave w issues pieces 4w .. 4w+3 (0..15: A's chunks, 16..31: B's)
static_assert(ACT_TILE_PTS % SLAB == 0, "a slab never straddles tiles");

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(4))) const u32x4 const_u32x4;

struct Frag { i32x2 q[4]; };                    // the lane's 32 bytes: q[0..1] = K block 0, q[2..3] = block 1
__device__ __forceinline__ void read_frag(Frag& f, unsigned addr) {
    asm volatile("ds_read_b64_tr_b8 %0, %4\n\tds_read_b64_tr_b8 %1, %4 offset:128\n\t"
                 "ds_read_b64_tr_b8 %2, %4 offset:512\n\tds_read_b64_tr_b8 %3, %4 offset:640"
                 : "=&v"(f.q[0]), "=&v"(f.q[1]), "=&v"(f.q[2]), "=&v"(f.q[3]) : "v"(addr));
}
__device__ __forceinline__ void frags_landed(Frag (&a)[4], Frag (&b)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0].q[0]), "+v"(a[0].q[1]), "+v"(a[0].q[2]), "+v"(a[0].q[3]), "+v"(a[1].q[0]), "+v"(a[1].q[1]),
                   "+v"(a[1].q[2]), "+v"(a[1].q[3]), "+v"(a[2].q[0]), "+v"(a[2].q[1]), "+v"(a[2].q[2]), "+v"(a[2].q[3]),
                   "+v"(a[3].q[0]), "+v"(a[3].q[1]), "+v"(a[3].q[2]), "+v"(a[3].q[3]), "+v"(b[0].q[0]), "+v"(b[0].q[1]),
                   "+v"(b[0].q[2]), "+v"(b[0].q[3]), "+v"(b[1].q[0]), "+v"(b[1].q[1]), "+v"(b[1].q[2]), "+v"(b[1].q[3]));
}
__device__ __forceinline__ i32x8 whole(const Frag& f) {
    return i32x8{f.q[0][0], f.q[0][1], f.q[1][0], f.q[1][1], f.q[2][0], f.q[2][1], f.q[3][0], f.q[3][1]};
}
// D = C in place (inline asm with a tied accumulator: with the builtin, sixteen-register accumulators that meet behind the
// wave-uniform tile branches were copied -- 64 spilled registers once the bias tile's accumulator came in).  The operands
// come from explicit waits (fragments) or from many instructions earlier (scales); the accumulators are read again one
// slab later, or behind the s_nops that follow the loop.
template <int OA, int OB>
__device__ __forceinline__ void mfma(const Frag& a, const Frag& b, f32x16& c, int sa, int sb) {
    const i32x8 av = whole(a), bv = whole(b);
    static_assert(OA >= 0 && OA < 4 && OB >= 0 && OB < 4, "op_sel picks one of four exponent bytes");
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[%5,%6,0] op_sel_hi:[%7,%8,0]"
                 : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb), "n"(OA & 1), "n"(OB & 1), "n"(OA >> 1), "n"(OB >> 1));
}

__global__ __launch_bounds__(512, 2) void dw_gemm_e4m3_kernel(GemmTableF8 tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int di = 0;
    while (di + 1 < tab.n && (int)blockIdx.x >= tab.d[di + 1].wg0) ++di;
    const GemmDescF8 d = tab.d[di];
    const int slice = blockIdx.x - d.wg0;
    const long long nslab = (tab.P + SLAB - 1) / SLAB;
    const long long s_begin = nslab * slice / d.wgs, s_end = nslab * (slice + 1) / d.wgs;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves over the 256 x 256 tile
    const int m0 = 128 * wm, n0 = 64 * wn;            // this wave: 4 x 2 tiles of 32 x 32
    const int M = d.A.width, N = d.B.width;

    // ---- LDS-DMA: piece = one chunk of the slab (64 points x 16 B, contiguous in HBM), lane l = point l
    const bool pieceA = wave < 4;
    const F8Operand op = pieceA ? d.A : d.B;
    const long long row_begin = s_begin * SLAB;
    const long long row_end = s_end * SLAB < tab.P ? s_end * SLAB : tab.P;
    const long long slice_rows = row_end > row_begin ? row_end - row_begin : 0;
    const long long tile0 = row_begin / ACT_TILE_PTS;
    const long long slice_tiles = (row_end + ACT_TILE_PTS - 1) / ACT_TILE_PTS - tile0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(op.data) + tile0 * op.block_bytes, 0, (int)(slice_rows > 0 ? slice_tiles * op.block_bytes : 0), 0x00020000);
    constexpr unsigned OUTSIDE = 0xffffffffu;
    unsigned voff[DMA_PER_SLAB], ldst[DMA_PER_SLAB];
#pragma unroll
    for (int k = 0; k < DMA_PER_SLAB; ++k) {
        const int piece = wave * DMA_PER_SLAB + k, c = piece & 15;
        ldst[k] = (piece >> 4) * OPB + c * CHS;
        voff[k] = c * 16 < op.width ? (unsigned)(c * (ACT_TILE_PTS * 16) + lane * 16) : OUTSIDE;
    }
    auto issue_slab = [&](int it, int slot) {
        const long long p0 = row_begin + (long long)it * SLAB;
        const unsigned vadd = (unsigned)((p0 / ACT_TILE_PTS - tile0) * op.block_bytes + (p0 % ACT_TILE_PTS) * 16);
        const int rows_left = tab.P - p0 < SLAB ? (int)(tab.P - p0) : SLAB;
#pragma unroll
        for (int k = 0; k < DMA_PER_SLAB; ++k) {
            lds_void* dst = reinterpret_cast<lds_void*>(reinterpret_cast<lds_char*>(0) + slot * SLOTB + ldst[k]);
            const unsigned off = (voff[k] == OUTSIDE || lane >= rows_left) ? OUTSIDE : voff[k] + vadd;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, off, 0, 0, 2);      // non-temporal: read once
        }
    };

    // ---- fragment addresses: lane (j = lane & 15, grp = (lane >> 4) & 1, h = lane >> 5) of tile t supplies, for the
    // transposing read, the 8 bytes (feature half j & 1) of point 16 h + (j >> 1) [+ 8, + 32, + 40 by the offsets] of
    // chunk 2 t + grp and receives feature 16 grp + j of the tile for 8 consecutive points per read
    const int h = lane >> 5;
    const unsigned lane_part = (unsigned)((16 * h + ((lane & 15) >> 1)) * 16 + (lane & 1) * 8 + ((lane >> 4) & 1) * CHS);
    unsigned aoff[4], boff[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)(2 * (m0 / 32 + i) * CHS) + lane_part;
#pragma unroll
    for (int j = 0; j < 2; ++j) boff[j] = OPB + (unsigned)(2 * (n0 / 32 + j) * CHS) + lane_part;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool active = m0 < M && n0 < N;
    // tiles of this wave that exist (wave-uniform): the others' exponent bytes are never looked at (amask / bmask)
    bool ta[4], tb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ta[i] = active && m0 + 32 * i < M;
#pragma unroll
    for (int j = 0; j < 2; ++j) tb[j] = active && n0 + 32 * j < N;
    // (the bias tile wn of a bias-only wave is among a's bytes too: its row tile exists whenever do_bias holds)
    unsigned amask = 0, bmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) amask |= (m0 + 32 * i < M) ? 0xffu << (8 * i) : 0u;
#pragma unroll
    for (int j = 0; j < 2; ++j) bmask |= (n0 + 32 * j < N) ? 0xffu << (8 * j) : 0u;

    // db = column sums of A = A^T . 1: ONE more MFMA per wave and slab against a fragment of ones (e4m3 1.0, exponent
    // 127).  Wave (wm, wn) takes row tile wn of its half: eight waves, eight row tiles, every row once.  (First version: two
    // 16-byte LDS reads and 48 conversions / adds per thread and slab, then a reduction through LDS.  A timing-only build
    // without them ran 12 % faster, but a same-box A/B of the two forms shows 1 %: 0.3074 -> 0.3042 ms for begin +
    // conversions + products.  Kept for having no vector work and no second LDS pass, not for the time.)
    const bool do_bias = d.bias && m0 + 32 * wn < M;
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const unsigned aoff_bias = (unsigned)(2 * (m0 / 32 + wn) * CHS) + lane_part;
    Frag ab, ones;
#pragma unroll
    for (int q = 0; q < 4; ++q) ab.q[q] = i32x2{0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) ones.q[q] = i32x2{0x38383838, 0x38383838};

    // exponents of the slab's two blocks: 16 contiguous bytes per operand
    auto load_scales = [&](long long s, u32x4& ea, u32x4& eb) {
        const long long blk = s * (SLAB / 32);
        ea = *reinterpret_cast<const_u32x4*>(reinterpret_cast<uintptr_t>(d.A.scales + blk * 8));
        eb = *reinterpret_cast<const_u32x4*>(reinterpret_cast<uintptr_t>(d.B.scales + blk * 8));
    };

#pragma unroll
    for (int k = 0; k < RING - 1; ++k)
        if (s_begin + k < s_end) issue_slab(k, k);
    u32x4 ea = {0, 0, 0, 0}, eb = {0, 0, 0, 0};
    if (s_begin < s_end) load_scales(s_begin, ea, eb);

    Frag a[4], b[2];
    for (long long s = s_begin; s < s_end; ++s) {
        const int it = (int)(s - s_begin), slot = it & (RING - 1);
        const long long left = s_end - s;
        const unsigned base = slot * SLOTB;
        // this wave's pieces of slab s have landed once at most the DMAs of the two younger slabs remain
        if (left >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // slab s published; everyone is done with slab s-1
        if (s + RING - 1 < s_end) issue_slab(it + RING - 1, (it + RING - 1) & (RING - 1));     // into the slot of slab s-1
        // the scale registers of this slab: lanes 0..31 carry block 0's exponent, lanes 32..63 block 1's
        const unsigned a_dw = (h ? (wm ? ea[3] : ea[2]) : (wm ? ea[1] : ea[0])) & amask;
        const unsigned b_dw = ((h ? ((wn >> 1) ? eb[3] : eb[2]) : ((wn >> 1) ? eb[1] : eb[0])) >> (16 * (wn & 1))) & bmask;
        const u32x4 ea_now = ea;
        if (left >= 2) load_scales(s + 1, ea, eb);
#if defined(F8_TIMING) && F8_TIMING == 4       // timing-only build: the LDS-DMA stream, its waits and barriers alone
        if (false) {
#else
        if (active || do_bias) {
#endif
            if (active) {
#pragma unroll
                for (int i = 0; i < 4; ++i) read_frag(a[i], base + aoff[i]);
#pragma unroll
                for (int j = 0; j < 2; ++j) read_frag(b[j], base + boff[j]);
            }
            // the bias tile's A fragment in registers of its own (row tile wn: a run-time index into a[] would be moves
            // or four code paths with the 16-register accumulator merged behind them -- 64 spills)
            if (do_bias) read_frag(ab, base + aoff_bias);
            frags_landed(a, b);
            asm volatile("" : "+v"(ab.q[0]), "+v"(ab.q[1]), "+v"(ab.q[2]), "+v"(ab.q[3]));     // (covered by the wait above)
            __builtin_amdgcn_sched_barrier(0);
            // (tiles past an operand's width are skipped; their exponent bytes are masked to 0 above all the same)
            if (ta[0] && tb[0]) mfma<0, 0>(a[0], b[0], acc[0][0], (int)a_dw, (int)b_dw);
            if (ta[0] && tb[1]) mfma<0, 1>(a[0], b[1], acc[0][1], (int)a_dw, (int)b_dw);
            if (ta[1] && tb[0]) mfma<1, 0>(a[1], b[0], acc[1][0], (int)a_dw, (int)b_dw);
            if (ta[1] && tb[1]) mfma<1, 1>(a[1], b[1], acc[1][1], (int)a_dw, (int)b_dw);
            if (ta[2] && tb[0]) mfma<2, 0>(a[2], b[0], acc[2][0], (int)a_dw, (int)b_dw);
            if (ta[2] && tb[1]) mfma<2, 1>(a[2], b[1], acc[2][1], (int)a_dw, (int)b_dw);
            if (ta[3] && tb[0]) mfma<3, 0>(a[3], b[0], acc[3][0], (int)a_dw, (int)b_dw);
            if (ta[3] && tb[1]) mfma<3, 1>(a[3], b[1], acc[3][1], (int)a_dw, (int)b_dw);
#if !(defined(F8_TIMING) && F8_TIMING == 5)   // timing-only build: no bias sums
            if (do_bias) {
                constexpr int ONE = 0x7f7f7f7f;        // exponent byte 127 = 2^0 for the ones, whatever op_sel picks
                mfma<0, 0>(ab, ones, accb, (int)(a_dw >> (8 * wn)), ONE);               // row tile wn's exponent into byte 0
            }
#endif
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (inline asm: no hazard recogniser) before they are read
    if (do_bias && s_begin < s_end && (lane & 31) == 0) {
        // every column of the tile holds the row sums: column 0 (lanes 0 and 32) adds them into the flat vector
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M) atomicAdd(d.bias + row, accb[r]);
        }
    }
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

// ---- producers of the narrow operands -------------------------------------------------------------------------------
// Row-major bf16 rows [P, ld] (columns 0 .. W-1, W in {16, 32, 64}) -> the 8-bit storage form of a W-wide operand.  One
// wave per (32-point block, 32-feature fragment): lane = point (lane & 31), 16-feature chunk (lane >> 5).
__global__ __launch_bounds__(256) void rows_to_e4m3_kernel(const __bf16* __restrict__ src, int ld, int W, long long P,
                                                           char* __restrict__ data, unsigned char* __restrict__ scales) {
    const int lane = threadIdx.x & 63;
    const int nq = (W + 31) / 32;
    const long long nblk = (P + 31) / 32;
    const long long wid = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (wid >= nblk * nq) return;
    const long long blk = wid / nq;
    const int Q = (int)(wid - blk * nq);
    const long long p = blk * 32 + (lane & 31);
    const int c = 2 * Q + (lane >> 5);                     // chunk
    const bool have = p < P && c * 16 < W;
    u32x4 w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
    if (have) {
        const u32x4* sp = reinterpret_cast<const u32x4*>(src + p * ld + c * 16);
        w0 = sp[0];
        w1 = sp[1];
    }
    // magnitudes of packed bf16 pairs order like their bit patterns with the sign cleared
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u16x2 x = __builtin_bit_cast(u16x2, w0[k] & 0x7fff7fffu), y = __builtin_bit_cast(u16x2, w1[k] & 0x7fff7fffu);
        m = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, m), __builtin_elementwise_max(x, y)));
    }
    unsigned mx = (m >> 16) > (m & 0xffffu) ? (m >> 16) : (m & 0xffffu);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)mx, off);
        mx = o > mx ? o : mx;
    }
    const int e = (int)(mx >> 7);
    const int sb = e > 8 ? e - 7 : 1;
    const float scale = __builtin_bit_cast(float, sb << 23);
    s16x2 t = {0, 0};
    auto cvt = [&](unsigned lo, unsigned hi) -> unsigned {
        t = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(t, __builtin_bit_cast(bf16x2, lo), scale, false);
        t = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(t, __builtin_bit_cast(bf16x2, hi), scale, true);
        return __builtin_bit_cast(unsigned, t);
    };
    const u32x4 o = {cvt(w0[0], w0[1]), cvt(w0[2], w0[3]), cvt(w1[0], w1[1]), cvt(w1[2], w1[3])};
    if (have)
        *reinterpret_cast<u32x4*>(data + (p / ACT_TILE_PTS) * f8_narrow_block_bytes(W) + ((long long)c * ACT_TILE_PTS + p % ACT_TILE_PTS) * 16) = o;
    if (lane == 0) scales[blk * 8 + Q] = (unsigned char)sb;
}

}  // namespace

// scratch_f8: the narrow operands in the 8-bit form, nerf_amd_f8_scratch_bytes(P): posx (64) | posd (32) | dsr (16)
static long long align256(long long v) { return (v + 255) / 256 * 256; }
extern "C" long long nerf_amd_f8_scratch_bytes(long long P) {
    return align256(f8_narrow_bytes(64, P)) + align256(f8_narrow_bytes(32, P)) + align256(f8_narrow_bytes(16, P));
}

// The narrow operands of the products -- the row-major bf16 encoder rows (which & 1) and the packed d_raw that
// nerf_amd_launch_param_gradients_begin leaves in `scratch` (which & 2) -- into `scratch_f8`.  Separate from the products so
// that a captured step can run each conversion beside the kernel that does not need it (the encoder rows beside the
// forward, d_raw beside the dX chain).
extern "C" int nerf_amd_launch_param_gradients_convert_e4m3(const void* posx64_v, const void* posd32_v, const void* scratch,
                                                            void* scratch_f8, long long P, int which, hipStream_t stream) {
    (void)hipGetLastError();
    if (P <= 0) return 0;
    char* px = reinterpret_cast<char*>(scratch_f8);
    char* pd = px + align256(f8_narrow_bytes(64, P));
    char* ds = pd + align256(f8_narrow_bytes(32, P));
    const long long nblk = (P + 31) / 32;
    auto convert = [&](const void* src, int ld, int W, char* dst) {
        const long long waves = nblk * ((W + 31) / 32);
        hipLaunchKernelGGL(rows_to_e4m3_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream,
                           reinterpret_cast<const __bf16*>(src), ld, W, P, dst,
                           reinterpret_cast<unsigned char*>(dst + f8_narrow_scale_offset(W, P)));
    };
    if (which & 1) {
        convert(posx64_v, 64, 64, px);
        convert(posd32_v, 32, 32, pd);
    }
    if (which & 2) convert(scratch, 32, 16, ds);
    return (int)hipGetLastError();
}

// The 8-bit counterpart of nerf_amd_launch_param_gradients_finish: acts / dys in the 8-bit storage form, the narrow operands
// already converted (above).
extern "C" int nerf_amd_launch_param_gradients_finish_e4m3(const void* acts_v, const void* dys_v, const void* scratch_f8,
                                                           float* grads, long long P, int bucket, hipStream_t stream) {
    (void)hipGetLastError();
    if (P <= 0) return 0;
    hipError_t e = hipSuccess;
    const char* acts = reinterpret_cast<const char*>(acts_v);
    const char* dys = reinterpret_cast<const char*>(dys_v);
    const char* px = reinterpret_cast<const char*>(scratch_f8);
    const char* pd = px + align256(f8_narrow_bytes(64, P));
    const char* ds = pd + align256(f8_narrow_bytes(32, P));

    auto blocked = [&](const char* buf, int L) {
        return F8Operand{buf + f8_offset_bytes(L, P), reinterpret_cast<const unsigned char*>(buf + f8_scale_offset_bytes(L, P)),
                         (int)F8_BLOCK_BYTES, act_width(L)};
    };
    auto narrow = [&](const char* buf, int W) {
        return F8Operand{buf, reinterpret_cast<const unsigned char*>(buf + f8_narrow_scale_offset(W, P)), (int)f8_narrow_block_bytes(W), W};
    };
    GemmTableF8 t{};
    t.P = P;
    int n = 0;
    auto add = [&](F8Operand A, F8Operand B, int coff, int ldc, int r0, int Mv, int Nv, int boff = -1) {
        GemmDescF8& g = t.d[n++];
        g.A = A; g.B = B; g.C = grads + coff; g.ldc = ldc; g.r0 = r0; g.Mv = Mv; g.Nv = Nv;
        g.bias = boff >= 0 ? grads + boff : nullptr;
    };
    const int LW = 256 * 256 + 256;
    add(blocked(dys, 0), narrow(px, 64), OFF_L0_W, 63, 0, 256, 63, OFF_L0_B);                               // layers_0.0
    for (int l = 1; l <= 4; ++l)                                                                             // layers_0.{2,4,6,8}
        add(blocked(dys, l), blocked(acts, l - 1), OFF_L1_W + (l - 1) * LW, 256, 0, 256, 256, OFF_L1_W + (l - 1) * LW + 65536);
    add(blocked(dys, 5), blocked(acts, 4), OFF_SKIP_W, 319, 0, 256, 256, OFF_SKIP_B);                        // skip [h ; x]: h part
    add(blocked(dys, 5), narrow(px, 64), OFF_SKIP_W + 256, 319, 0, 256, 63);                                //               x part
    add(blocked(dys, 6), blocked(acts, 5), OFF_L6_W, 256, 0, 256, 256, OFF_L6_W + 65536);                    // layers_1.0
    add(blocked(dys, 7), blocked(acts, 6), OFF_L6_W + LW, 256, 0, 256, 256, OFF_L6_W + LW + 65536);          // layers_1.2
    add(narrow(ds, 16), blocked(acts, 7), OFF_SIG_W, 256, 3, 1, 256);                                        // sigma_fc.0 (row 3 of dsr)
    add(blocked(dys, 8), blocked(acts, 7), OFF_L2_W, 256, 0, 256, 256, OFF_L2_B);                            // layers_2
    add(blocked(dys, 9), blocked(acts, 8), OFF_C0_W, 283, 0, 128, 256, OFF_C0_B);                            // color_fc.0 [h ; d]: h part
    add(blocked(dys, 9), narrow(pd, 32), OFF_C0_W + 256, 283, 0, 128, 27);                                   //                      d part
    add(narrow(ds, 16), blocked(acts, 9), OFF_C1_W, 128, 0, 3, 128);                                         // color_fc.2 (rows 0..2)
    if (bucket != 0) {
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const bool head = t.d[i].C - grads < GRAD_BUCKET_SPLIT;
            if (head == (bucket == 2)) t.d[m++] = t.d[i];
        }
        n = m;
    }
    t.n = n;
    // workgroup shares as in dw_gemm.hip: bytes streamed per point (M + N) plus a fixed cost per slab
    constexpr double slab_cost = 1024.0;
    const int cus = device_cus();
    double total = 0;
    for (int i = 0; i < n; ++i) total += (double)(t.d[i].A.width + t.d[i].B.width) + slab_cost;
    const long long nslab = (P + SLAB - 1) / SLAB;
    const long long need = (P * 256 >> 30) + 1;            // a slice's operand stays below 1 GiB (32-bit buffer range)
    long long w[16];
    double rem[16];
    long long used = 0;
    for (int i = 0; i < n; ++i) {
        const double share = ((double)(t.d[i].A.width + t.d[i].B.width) + slab_cost) / total * cus;
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
        ++w[best]; rem[best] = 0; ++used;
    }
    int wg = 0;
    for (int i = 0; i < n; ++i) {
        t.d[i].wg0 = wg;
        t.d[i].wgs = (int)w[i];
        wg += (int)w[i];
    }
    e = allow_dynamic_lds(reinterpret_cast<const void*>(dw_gemm_e4m3_kernel), LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dw_gemm_e4m3_kernel, dim3(wg), dim3(512), LDS_BYTES, stream, t);
    return (int)hipGetLastError();
}
