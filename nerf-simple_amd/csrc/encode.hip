// encode.hip -- stand-alone positional encoding kernels, the drop-in bodies of
// gamma (reference utils/xyz.py:6-14) and positional_encoder (utils/xyz.py:16-36).
// The fused MLP kernels encode in registers and never call these; they exist
// for callers that want posx/posd themselves.
//
// HBM-bound: 24 B read and (3+6Lp + 3+6Ld)*4 = 360 B written per point.  One
// thread per OUTPUT element so stores are fully coalesced; the six inputs of a
// point are re-read through L1/L2.  sin/cos use the accurate ocml sinf/cosf on
// the exactly-scaled argument 2^i * x (a power-of-two product is exact in fp32,
// as it is in the reference), so results track torch-CPU to ~1 ulp.
#include "nerf_device.h"

namespace {

__device__ __forceinline__ float enc_value(float x, int idx) {
    // idx = 2*level + trig
    const float a = ldexpf(x, idx >> 1);
    return (idx & 1) ? cosf(a) : sinf(a);
}

// out[n, 2L]
__global__ void gamma_kernel(const float* __restrict__ x, long long x_stride,
                             float* __restrict__ out, long long n, int L) {
    const long long total = n * 2 * L;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long p = e / (2 * L);
        const int idx = (int)(e - p * 2 * L);
        out[e] = enc_value(x[p * x_stride], idx);
    }
}

// one launch writes both tables: columns [0, Cx) -> posx, [Cx, Cx+Cd) -> posd
__global__ void posenc_kernel(const float* __restrict__ vec, float* __restrict__ posx,
                              float* __restrict__ posd, long long P, int Lp, int Ld) {
    const int Cx = 3 + 6 * Lp, Cd = 3 + 6 * Ld, C = Cx + Cd;
    const long long total = P * C;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long p = e / C;
        int col = (int)(e - p * C);
        const float* v = vec + p * 6;
        float* dst;
        int L, base;
        if (col < Cx) { dst = posx + p * Cx + col; L = Lp; base = 0; }
        else { col -= Cx; dst = posd + p * Cd + col; L = Ld; base = 3; }
        float r;
        if (col < 3) {
            r = v[base + col];
        } else {
            const int c = (col - 3) / (2 * L);
            r = enc_value(v[base + c], (col - 3) - c * 2 * L);
        }
        *dst = r;
    }
}

// Training-side front end: sampling + point assembly + encoding in one pass
// (reference utils/rendering.py:24-40 + utils/xyz.py:16-36), writing the
// encoder outputs the dense layers consume: posx[P,63], posd[P,27], ts[B,N].
// One thread per output element of the concatenated [63 | 27] row.
__global__ void sample_encode_kernel(MlpArgs a, float* __restrict__ posx, float* __restrict__ posd) {
    constexpr int Cx = 63, Cd = 27, C = Cx + Cd;
    const long long total = a.P * C;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long p = e / C;
        int col = (int)(e - p * C);
        const PointIn pt = fetch_point_rays(a, p);
        if (col == 0 && a.ts_out) a.ts_out[p] = pt.t;
        const float xyz[3] = {pt.x, pt.y, pt.z}, dd[3] = {pt.d1, pt.d2, pt.d3};
        if (col < Cx) {
            posx[p * Cx + col] = col < 3 ? xyz[col] : enc_value(xyz[(col - 3) / 20], (col - 3) % 20);
        } else {
            col -= Cx;
            posd[p * Cd + col] = col < 3 ? dd[col] : enc_value(dd[(col - 3) / 8], (col - 3) % 8);
        }
    }
}

// The same front end for the fused training path: bf16 outputs padded to the dW GEMM's
// operand widths, posx [P,64] (col 63 = 0) and posd [P,32] (cols 27..31 = 0).
// One thread per point: the point is fetched once, the 84 sines / cosines come from the
// hardware v_sin / v_cos on arguments in revolutions (to_revolutions + sincos_rev_fast, the
// encoder of the bf16 MLP kernels, ~1e-6 abs -- far inside bf16), and each thread writes its two
// rows as 16-byte stores.
typedef __bf16 enc_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void sample_encode_bf16_kernel(MlpArgs a, __bf16* __restrict__ posx,
                                                                 __bf16* __restrict__ posd) {
    const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (p >= a.P) return;
    const PointIn pt = a.pts ? fetch_point_pts(a, p) : fetch_point_rays(a, p);      // points mode: Nerf.forward(v)
    if (a.ts_out) a.ts_out[p] = pt.t;
    {
        __bf16 row[64];
        const float xyz[3] = {pt.x, pt.y, pt.z};
#pragma unroll
        for (int cd = 0; cd < 3; ++cd) {
            row[cd] = (__bf16)xyz[cd];
            const TwoF q = to_revolutions(xyz[cd]);
#pragma unroll
            for (int l = 0; l < 10; ++l) {
                float sn, cs;
                sincos_rev_fast(q, __builtin_amdgcn_ldexpf(1.0f, l), sn, cs);
                row[3 + 20 * cd + 2 * l] = (__bf16)sn;
                row[3 + 20 * cd + 2 * l + 1] = (__bf16)cs;
            }
        }
        row[63] = (__bf16)0.f;
        enc_bf16x8* dst = reinterpret_cast<enc_bf16x8*>(posx + p * 64);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            dst[k] = enc_bf16x8{row[8 * k], row[8 * k + 1], row[8 * k + 2], row[8 * k + 3],
                                row[8 * k + 4], row[8 * k + 5], row[8 * k + 6], row[8 * k + 7]};
    }
    {
        __bf16 row[32];
        const float dd[3] = {pt.d1, pt.d2, pt.d3};
#pragma unroll
        for (int cd = 0; cd < 3; ++cd) {
            row[cd] = (__bf16)dd[cd];
            const TwoF q = to_revolutions(dd[cd]);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                float sn, cs;
                sincos_rev_fast(q, __builtin_amdgcn_ldexpf(1.0f, l), sn, cs);
                row[3 + 8 * cd + 2 * l] = (__bf16)sn;
                row[3 + 8 * cd + 2 * l + 1] = (__bf16)cs;
            }
        }
#pragma unroll
        for (int k = 27; k < 32; ++k) row[k] = (__bf16)0.f;
        enc_bf16x8* dst = reinterpret_cast<enc_bf16x8*>(posd + p * 32);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            dst[k] = enc_bf16x8{row[8 * k], row[8 * k + 1], row[8 * k + 2], row[8 * k + 3],
                                row[8 * k + 4], row[8 * k + 5], row[8 * k + 6], row[8 * k + 7]};
    }
}

// Sampling + query-point assembly on their own (reference utils/rendering.py:24-40), for callers whose network is
// not the fused one: query_pts[P,6] = [o + d t, d / ||d||] ray-major / sample-minor, ts[B,N].  One thread per point.
__global__ __launch_bounds__(256) void query_points_kernel(MlpArgs a, float* __restrict__ query_pts) {
    const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (p >= a.P) return;
    const PointIn pt = fetch_point_rays(a, p);
    if (a.ts_out) a.ts_out[p] = pt.t;
    float* q = query_pts + p * 6;
    q[0] = pt.x; q[1] = pt.y; q[2] = pt.z; q[3] = pt.d1; q[4] = pt.d2; q[5] = pt.d3;
}

// The reference's range check (utils/xyz.py:8-9: `if torch.any(x < -1) or torch.any(x > 1): warnings.warn(...)` in every
// gamma call of positional_encoder, i.e. on all six columns of the query points) without materialising the points and
// without a host sync: *word |= 1 if any coordinate of any query point lies outside [-1, 1].
//   rays mode (a.rays): one thread per RAY.  A coordinate o + d t is monotone in t -- also as the kernels round it: one
//   rounded product, one rounded sum -- so its extremes over a ray's samples sit at the first and the last sample:
//   two of the ray's N points decide for all of them, and those two are formed exactly as the render forms them
//   (fetch_point_rays: explicit jitter, explicit positions or the counter RNG).  The unit direction is checked too.
//   explicit positions (NERF_FLAG_TS_GIVEN) need not be sorted: every sample of the ray is looked at.
//   flat mode (a.pts): B floats -- the six columns of given points, or any gamma() argument -- one thread per float.
__global__ __launch_bounds__(256) void range_check_kernel(MlpArgs a, long long B, unsigned* __restrict__ word) {
    const long long b = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    bool out = false;
    if (a.pts) {
        if (b < B) {
            const float v = a.pts[b];
            out = v < -1.f || v > 1.f;          // torch.any(x < -1) or torch.any(x > 1): NaN compares false and does not warn
        }
    } else if (b < B) {
        const bool all = (a.flags & NERF_FLAG_TS_GIVEN) != 0;
        const int step = all || a.N == 1 ? 1 : a.N - 1;
        for (int i = 0; i < a.N; i += step) {
            const PointIn pt = fetch_point_rays(a, b * a.N + i, RaySample{b, i});
            out |= pt.x < -1.f || pt.x > 1.f || pt.y < -1.f || pt.y > 1.f || pt.z < -1.f || pt.z > 1.f || pt.d1 < -1.f ||
                   pt.d1 > 1.f || pt.d2 < -1.f || pt.d2 > 1.f || pt.d3 < -1.f || pt.d3 > 1.f;
        }
    }
    if (__any(out) && (threadIdx.x & 63) == 0) atomicOr(word, 1u);       // at most one atomic per wave
}

__host__ int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}

}  // namespace

extern "C" int nerf_amd_launch_gamma(const float* x, long long x_stride, float* out, long long n,
                                     int L, hipStream_t stream) {
    (void)hipGetLastError();   // drop any stale error: the return value is about THIS launch
    if (n == 0 || L == 0) return 0;
    hipLaunchKernelGGL(gamma_kernel, dim3(grid_for(n * 2 * L)), dim3(256), 0, stream, x, x_stride, out, n, L);
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_posenc(const float* vec, float* posx, float* posd, long long P,
                                      int Lp, int Ld, hipStream_t stream) {
    (void)hipGetLastError();   // drop any stale error: the return value is about THIS launch
    if (P == 0) return 0;
    hipLaunchKernelGGL(posenc_kernel, dim3(grid_for(P * (6 + 6 * Lp + 6 * Ld))), dim3(256), 0, stream,
                       vec, posx, posd, P, Lp, Ld);
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_sample_encode(const MlpArgs* args, float* posx, float* posd, hipStream_t stream) {
    (void)hipGetLastError();
    if (args->P == 0) return 0;
    hipLaunchKernelGGL(sample_encode_kernel, dim3(grid_for(args->P * 90)), dim3(256), 0, stream, *args, posx, posd);
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_query_points(const MlpArgs* args, float* query_pts, hipStream_t stream) {
    (void)hipGetLastError();
    if (args->P == 0) return 0;
    hipLaunchKernelGGL(query_points_kernel, dim3((unsigned)((args->P + 255) / 256)), dim3(256), 0, stream, *args, query_pts);
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_range_check(const MlpArgs* args, long long B, unsigned* word, hipStream_t stream) {
    (void)hipGetLastError();
    const long long n = B;
    if (n == 0) return 0;
    hipLaunchKernelGGL(range_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, *args, B, word);
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_sample_encode_bf16(const MlpArgs* args, void* posx64, void* posd32, hipStream_t stream) {
    (void)hipGetLastError();
    if (args->P == 0) return 0;
    hipLaunchKernelGGL(sample_encode_bf16_kernel, dim3((unsigned)((args->P + 255) / 256)), dim3(256), 0, stream, *args,
                       reinterpret_cast<__bf16*>(posx64), reinterpret_cast<__bf16*>(posd32));
    return (int)hipGetLastError();
}
