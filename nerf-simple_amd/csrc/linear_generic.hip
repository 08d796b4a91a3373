// linear_generic.hip -- one strided fp32 GEMM on v_mfma_f32_16x16x4_f32 for the networks the fused kernels are NOT
// built for: the reference's Nerf(Lp, Ld, H) takes any sizes (utils/nets.py:9-32), the fused kernels implement the one
// it ever constructs (Nerf() = (10, 4, 256), train.py:41, test.py:27).  Every nn.Linear of such a network -- forward
// (y = x W^T + b, optional ReLU), and in the backward dX = dY W, dW = dY^T X, db = dY^T 1 -- is this kernel with other
// strides (utils/generic_mlp.py lists the calls):
//
//     C[i, j]  (+)=  sum_k  A(i, k) * B(k, j)  (+ bias[j])  (ReLU)          i < M, j < N, k < K
//     A(i, k) = A[i * sa_i + k * sa_k]   -- or 0 where A_mask[i * sa_i + k * sa_k] <= 0 (the ReLU derivative of a
//                                           saved activation, applied to the incoming gradient on the fly)
//     B(k, j) = B[k * sb_k + j * sb_j]   -- a weight matrix, a slice of one (concatenated inputs), a saved activation,
//                                           or a single 1.0f with both strides 0 (column sums = bias gradients)
//
// Exact fp32 products and fp32 accumulation (the MFMA is an fma chain); a correct, tidy kernel rather than a tuned
// one: 64 x 64 output tile per workgroup of four waves, K in steps of 16 through LDS (the next step's operands are
// fetched into registers under the current step's MFMAs), loads coalesced along whichever index is contiguous.  Long reductions with few output tiles (dW: K = number of points) are split over blockIdx.z and
// summed with float atomics.
#include "nerf_device.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;
constexpr int LIN_RELU = 1, LIN_ACCUMULATE = 2;

struct LinArgs {
    const float* A;
    long long sa_i, sa_k;
    const float* A_mask;
    const float* B;
    long long sb_k, sb_j;
    const float* bias;
    float* C;
    long long ldc;
    long long M, N, K;
    long long k_chunk;          // K range of one blockIdx.z
    int flags;
    int atomic;                 // split-K: add the partial sums atomically
};

__global__ __launch_bounds__(256) void linear_f32_kernel(LinArgs a) {
    // [row / column][k] with an odd row stride (17 words): the operand stores -- along k or along the row index, whichever
    // is contiguous in memory -- and the fragment reads (16 rows x 4 k per wave-instruction) all touch 32 distinct banks
    __shared__ float As[TM][TK + 1];
    __shared__ float Bs[TN][TK + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long i0 = (long long)blockIdx.x * TM, j0 = (long long)blockIdx.y * TN;
    const long long k_lo = (long long)blockIdx.z * a.k_chunk;
    const long long k_hi = k_lo + a.k_chunk < a.K ? k_lo + a.k_chunk : a.K;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;      // this wave's 32 x 32 quarter of the tile
    f32x4 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool a_i_fast = a.sa_i == 1 && a.sa_k != 1;            // consecutive threads along the contiguous index
    const bool b_j_fast = a.sb_j == 1 || a.sb_k != 1;
    // this thread's four elements of each operand tile: (row / column, k) inside the tile
    constexpr int RA = TM * TK / 256, RB = TN * TK / 256;
    int ai[RA], ak[RA], bj[RB], bk[RB];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int e = tid + 256 * r;
        ai[r] = a_i_fast ? e % TM : e / TK;
        ak[r] = a_i_fast ? e / TM : e % TK;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int e = tid + 256 * r;
        bj[r] = b_j_fast ? e % TN : e / TK;
        bk[r] = b_j_fast ? e / TN : e % TK;
    }
    float ra[RA], rb[RB];
    auto fetch = [&](long long k0) {              // global -> registers (zeros outside the matrices / this K range)
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            float v = 0.f;
            if (i0 + ai[r] < a.M && k0 + ak[r] < k_hi) {
                const long long off = (i0 + ai[r]) * a.sa_i + (k0 + ak[r]) * a.sa_k;
                v = a.A[off];
                if (a.A_mask && a.A_mask[off] <= 0.f) v = 0.f;      // torch threshold_backward: a NaN activation lets the gradient through
            }
            ra[r] = v;
        }
#pragma unroll
        for (int r = 0; r < RB; ++r)
            rb[r] = (j0 + bj[r] < a.N && k0 + bk[r] < k_hi) ? a.B[(k0 + bk[r]) * a.sb_k + (j0 + bj[r]) * a.sb_j] : 0.f;
    };
    if (k_lo < k_hi) fetch(k_lo);
    for (long long k0 = k_lo; k0 < k_hi; k0 += TK) {
#pragma unroll
        for (int r = 0; r < RA; ++r) As[ai[r]][ak[r]] = ra[r];
#pragma unroll
        for (int r = 0; r < RB; ++r) Bs[bj[r]][bk[r]] = rb[r];
        __syncthreads();
        if (k0 + TK < k_hi) fetch(k0 + TK);        // the next step's loads fly under this step's MFMAs
#pragma unroll
        for (int ks = 0; ks < TK / 4; ++ks) {
            const int kk = 4 * ks + (lane >> 4);
            float af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = As[wi + 16 * t + (lane & 15)][kk];
                bf[t] = Bs[wj + 16 * t + (lane & 15)][kk];
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ti], bf[tj], acc[ti][tj], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[m = 4 * (lane / 16) + r][n = lane % 16] per 16 x 16 tile
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long i = i0 + wi + 16 * ti + 4 * (lane >> 4) + r, j = j0 + wj + 16 * tj + (lane & 15);
                if (i >= a.M || j >= a.N) continue;
                float* dst = a.C + i * a.ldc + j;
                float v = acc[ti][tj][r];
                if (a.atomic) {
                    atomicAdd(dst, v);
                    continue;
                }
                if (a.bias) v += a.bias[j];
                if (a.flags & LIN_ACCUMULATE) v += *dst;
                if (a.flags & LIN_RELU) v = v < 0.f ? 0.f : v;       // keeps NaN, as torch's relu does
                *dst = v;
            }
}

// The bias gradients -- B a single 1.0f with both strides 0, N = 1, A read along its contiguous index (sa_i == 1) -- are
// column sums, not a GEMM: one thread per output row, coalesced loads, K split over blockIdx.y, float atomics.
// (As a 64-wide MFMA tile with one live column they took a quarter of the exact training step.)
__global__ __launch_bounds__(256) void colsum_f32_kernel(LinArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.M) return;
    const long long k_lo = (long long)blockIdx.y * a.k_chunk;
    const long long k_hi = k_lo + a.k_chunk < a.K ? k_lo + a.k_chunk : a.K;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long long k = k_lo;
    for (; k + 4 <= k_hi; k += 4) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long off = i + (k + r) * a.sa_k;
            v[r] = a.A[off];
            if (a.A_mask && a.A_mask[off] <= 0.f) v[r] = 0.f;
        }
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
    }
    for (; k < k_hi; ++k) {
        const long long off = i + k * a.sa_k;
        float v = a.A[off];
        if (a.A_mask && a.A_mask[off] <= 0.f) v = 0.f;      // torch threshold_backward: a NaN activation lets the gradient through
        s0 += v;
    }
    atomicAdd(a.C + i * a.ldc, ((s0 + s1) + (s2 + s3)) * a.B[0]);
}

}  // namespace

extern "C" int nerf_amd_launch_linear_f32(const float* A, long long sa_i, long long sa_k, const float* A_mask, const float* B,
                                          long long sb_k, long long sb_j, const float* bias, float* C, long long ldc,
                                          long long M, long long N, long long K, int flags, hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || N <= 0) return 0;
    LinArgs a{A, sa_i, sa_k, A_mask, B, sb_k, sb_j, bias, C, ldc, M, N, K, K > 0 ? K : 1, flags, 0};
    if (N == 1 && sb_k == 0 && sb_j == 0 && sa_i == 1 && K > 0 && flags == LIN_ACCUMULATE && !bias) {
        a.k_chunk = (K + 32767) / 32768 > 256 ? (K + 32767) / 32768 : 256;      // many short chunks: enough loads in flight
        const long long gy = (K + a.k_chunk - 1) / a.k_chunk, gxs = (M + 255) / 256;
        if (gy > 65535 || gxs > 2147483647ll) return -1;
        hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)gxs, (unsigned)gy), dim3(256), 0, stream, a);
        return (int)hipGetLastError();
    }
    const long long gx = (M + TM - 1) / TM, gy = (N + TN - 1) / TN;
    long long gz = 1;
    // few output tiles and a long reduction (weight / bias gradients over all points): split K, sum with atomics
    if ((flags & LIN_ACCUMULATE) && !(flags & LIN_RELU) && !bias && K >= 8192 && gx * gy < 1024) {
        gz = (K + 4095) / 4096;
        const long long want = 2048 / (gx * gy);
        if (gz > want) gz = want;
        if (gz < 1) gz = 1;
    }
    if (gz > 1) {
        a.k_chunk = ((K + gz - 1) / gz + TK - 1) / TK * TK;
        gz = (K + a.k_chunk - 1) / a.k_chunk;
        a.atomic = 1;
    }
    if (gx > 2147483647ll || gy > 65535 || gz > 65535) return -1;
    hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)gz), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}
