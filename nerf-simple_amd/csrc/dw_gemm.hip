// dw_gemm.hip -- parameter gradients of the dense layers (training step,
// reference train.py:51-54: loss.backward() as far as the 24 parameter tensors).
//
//   dW_l = dY_l^T @ X_l      (sum over all P = B*N query points)
//   db_l = sum_p dY_l[p, :]  (accumulated from the A operand's staging registers of the same kernel)
//
// dY_l (from nerf_amd_mlp_backward) and X_l (the activations saved by
// nerf_amd_mlp_forward_train, plus the encoder outputs) are [P, width] bf16
// ROW-major, so both MFMA operands have the reduction index (the point) as
// their slow dimension: a "TN" GEMM with M, N <= 256 and K = P ~ 10^5..10^6.
// The vendor library runs this shape on 16 workgroups; here:
//   * ONE launch covers all 14 products; each gets a share of the ~256
//     workgroups proportional to the bytes it streams (split-K over the points);
//   * a workgroup (8 waves) owns the full 256x256 output of its product in
//     registers and walks its K slice in slabs of 64 points, staged
//     HBM -> registers -> LDS (row-major, rows padded to 576 B), double buffered;
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
    const __bf16* A;   // [P, lda], columns 0..M-1 become output ROWS
    const __bf16* B;   // [P, ldb], columns 0..N-1 become output COLUMNS
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

constexpr int SLAB = 64;                       // points per LDS slab (4 k-steps of 16)
constexpr int ROWB = 576;                      // LDS row stride: 512 B of features + 64 B
                                               // (rows 16 banks apart: conflict-free tr reads)
constexpr int OPB = SLAB * ROWB;               // one operand slab = 36 KiB
constexpr int LDS_BYTES = 4 * OPB;             // {A,B} x 2 buffers = 144 KiB

typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ bf16x8 read_frag_tr(unsigned addr) {
    // two transposing reads: k = 8h + 0..3 and 8h + 4..7 of this lane's column
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(reinterpret_cast<lds_char*>(0) + addr));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(reinterpret_cast<lds_char*>(0) + addr + 4 * ROWB));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
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

    // staging geometry: thread -> (row tid>>5 + 16 i, 16-B chunk tid&31) of a slab, i = 0..3
    const int srow = tid >> 5, schunk = tid & 31;
    const bool a_col = schunk * 8 < d.M, b_col = schunk * 8 < d.N;
    u32x4 ra[4], rb[4];
    // db = column sums of dY: this thread sees 8 columns x 4 rows of A per slab in its staging registers
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto add_bias = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, ra[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) bsum[k] += (float)v[k];
        }
    };
    auto load_slab = [&](long long s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long p = s * SLAB + srow + 16 * i;
            ra[i] = u32x4{0u, 0u, 0u, 0u};
            rb[i] = u32x4{0u, 0u, 0u, 0u};
            if (p < tab.P) {
                if (a_col) ra[i] = *reinterpret_cast<const u32x4*>(d.A + p * d.lda + schunk * 8);
                if (b_col) rb[i] = *reinterpret_cast<const u32x4*>(d.B + p * d.ldb + schunk * 8);
            }
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned off = (srow + 16 * i) * ROWB + schunk * 16;
            *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(
                reinterpret_cast<lds_char*>(0) + buf * 2 * OPB + off) = ra[i];
            *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(
                reinterpret_cast<lds_char*>(0) + buf * 2 * OPB + OPB + off) = rb[i];
        }
    };
    // fragment address of this lane inside a slab: row (8h + q), column 16 (group&1) + 4p
    const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = lane >> 5;
    const unsigned frag_off = (8 * h + q) * ROWB + (16 * (grp & 1) + 4 * pp) * 2;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // waves whose tile rows / columns lie outside the product skip the arithmetic
    const bool active = m0 < d.M && n0 < d.N;

    if (s_begin < s_end) {
        load_slab(s_begin);
        if (d.bias) add_bias();
        store_slab(0);
    }
    __syncthreads();
    for (long long s = s_begin; s < s_end; ++s) {
        const int buf = (int)((s - s_begin) & 1);
        if (s + 1 < s_end) load_slab(s + 1);
        if (active) {
            const unsigned abase = buf * 2 * OPB + frag_off, bbase = abase + OPB;
#pragma unroll
            for (int ks = 0; ks < SLAB / 16; ++ks) {
                bf16x8 af[4], bf[2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    af[i] = read_frag_tr(abase + ks * 16 * ROWB + (m0 + 32 * i) * 2);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bf[j] = read_frag_tr(bbase + ks * 16 * ROWB + (n0 + 32 * j) * 2);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (s + 1 < s_end) {
            if (d.bias) add_bias();
            store_slab(buf ^ 1);
        }
        __syncthreads();
    }
    if (d.bias && a_col && s_begin < s_end) {
        // 16 threads (srow) hold partial sums of the same 8 columns: combine through LDS, then atomics
        float* red = reinterpret_cast<float*>(smem);              // all slab reads are behind the last barrier
#pragma unroll
        for (int k = 0; k < 8; ++k) red[(srow * 32 + schunk) * 8 + k] = bsum[k];
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

}  // namespace

// grads: flat fp32 [595844] in state_dict order (zeroed here); scratch: P*64 bytes (dsr)
extern "C" int nerf_amd_launch_param_gradients(const float* d_raw, const void* acts_v, const void* dys_v,
                                               const void* posx64_v, const void* posd32_v, void* scratch,
                                               float* grads, long long P, hipStream_t stream) {
    (void)hipGetLastError();
    hipError_t e = hipMemsetAsync(grads, 0, sizeof(float) * PARAM_COUNT, stream);
    if (e != hipSuccess) return (int)e;
    if (P <= 0) return 0;
    const __bf16* acts = reinterpret_cast<const __bf16*>(acts_v);
    const __bf16* dys = reinterpret_cast<const __bf16*>(dys_v);
    const __bf16* posx = reinterpret_cast<const __bf16*>(posx64_v);
    const __bf16* posd = reinterpret_cast<const __bf16*>(posd32_v);
    __bf16* dsr = reinterpret_cast<__bf16*>(scratch);
    auto act = [&](int L) { return acts + (long long)L * P * 256; };      // bf16 elements: L*P*512 bytes
    auto dy = [&](int L) { return dys + (long long)L * P * 256; };

    hipLaunchKernelGGL(pack_draw_kernel, dim3(512), dim3(256), 0, stream, d_raw, dsr, P, grads + OFF_C1_B,
                       grads + OFF_SIG_B);

    GemmTable t{};
    t.P = P;
    int n = 0;
    auto add = [&](const __bf16* A, int lda, int M, const __bf16* B, int ldb, int N, int coff, int ldc, int r0,
                   int Mv, int Nv, int boff = -1) {
        GemmDesc& g = t.d[n++];
        g.A = A; g.lda = lda; g.M = M; g.B = B; g.ldb = ldb; g.N = N;
        g.C = grads + coff; g.ldc = ldc; g.r0 = r0; g.Mv = Mv; g.Nv = Nv;
        g.bias = boff >= 0 ? grads + boff : nullptr;      // db of the layer whose dY is this product's A
    };
    const int LW = 256 * 256 + 256;
    add(dy(0), 256, 256, posx, 64, 64, OFF_L0_W, 63, 0, 256, 63, OFF_L0_B);               // layers_0.0
    for (int l = 1; l <= 4; ++l)                                                          // layers_0.{2,4,6,8}
        add(dy(l), 256, 256, act(l - 1), 256, 256, OFF_L1_W + (l - 1) * LW, 256, 0, 256, 256,
            OFF_L1_W + (l - 1) * LW + 65536);
    add(dy(5), 256, 256, act(4), 256, 256, OFF_SKIP_W, 319, 0, 256, 256, OFF_SKIP_B);     // skip [h ; x]: h part
    add(dy(5), 256, 256, posx, 64, 64, OFF_SKIP_W + 256, 319, 0, 256, 63);                //               x part
    add(dy(6), 256, 256, act(5), 256, 256, OFF_L6_W, 256, 0, 256, 256, OFF_L6_W + 65536); // layers_1.0
    add(dy(7), 256, 256, act(6), 256, 256, OFF_L6_W + LW, 256, 0, 256, 256, OFF_L6_W + LW + 65536);  // layers_1.2
    add(dsr, 32, 32, act(7), 256, 256, OFF_SIG_W, 256, 3, 1, 256);                        // sigma_fc.0 (row 3 of dsr)
    add(dy(8), 256, 256, act(7), 256, 256, OFF_L2_W, 256, 0, 256, 256, OFF_L2_B);         // layers_2
    add(dy(9), 128, 128, act(8), 256, 256, OFF_C0_W, 283, 0, 128, 256, OFF_C0_B);         // color_fc.0 [h ; d]: h part
    add(dy(9), 128, 128, posd, 32, 32, OFF_C0_W + 256, 283, 0, 128, 27);                  //                      d part
    add(dsr, 32, 32, act(9), 128, 128, OFF_C1_W, 128, 0, 3, 128);                         // color_fc.2 (rows 0..2)
    t.n = n;
    // workgroups per product, about one per CU in total.  The kernel is HBM-bound, so a product's
    // cost per slab is the bytes it streams, (M + N) * 2 per point, not its M*N flops (sizing by
    // flops left the thin products -- 32 x 256 reads as much of X as 256 x 256 -- as a 2 ms tail)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    double total = 0;
    for (int i = 0; i < n; ++i) total += (double)(t.d[i].M + t.d[i].N);
    const long long nslab = (P + SLAB - 1) / SLAB;
    int wg = 0;
    for (int i = 0; i < n; ++i) {
        long long w = (long long)((double)(t.d[i].M + t.d[i].N) / total * cus + 0.5);
        if (w < 1) w = 1;
        if (w > nslab) w = nslab;
        t.d[i].wg0 = wg;
        t.d[i].wgs = (int)w;
        wg += (int)w;
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(dw_gemm_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dw_gemm_kernel, dim3(wg), dim3(512), LDS_BYTES, stream, t);

    return (int)hipGetLastError();
}
