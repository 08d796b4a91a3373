// composite.hip -- sigma -> alpha compositing along each ray: the drop-in body
// of volume_render (reference utils/rendering.py:47-85).
//
// One wavefront per ray, one sample per lane, N walked in chunks of 64 with a
// running transmittance carried between chunks.  The reference's exclusive
// cumulative PRODUCT of (1 - alpha + 1e-10) (utils/rendering.py:68; not
// exp(-cumsum)) is a wave-level inclusive product scan (6 shuffle steps)
// shifted by one lane.  fp32 throughout, literal formulas: softplus(beta=1,
// threshold=20), last delta = 1e10, deltas scaled by ||dirs||, second output is
// DISPARITY 1/max(1e-10, depth/acc) with torch.max's NaN propagation (acc == 0
// gives NaN exactly like the reference).
//
// 20 B read per sample (+8 B written when alpha / w are requested) and 20 B written per ray.
// Not HBM-bound in practice: the exact-fp32 softplus / exp (ocml expf, log1pf) and the two scans
// cost ~400 VALU instructions per 64 samples, which is what sets its 3.4 TB/s.
#include "composite_device.h"

namespace {

using nerf_composite::wave_sum;

constexpr int RAYS_PER_BLOCK = 4;

struct GlobalSamples {                        // the ray's samples in HBM
    const float* rts;
    const f32x4* rraw;
    __device__ __forceinline__ float t(int i) const { return rts[i]; }
    __device__ __forceinline__ f32x4 c(int i) const { return rraw[i]; }
};

__global__ __launch_bounds__(64 * RAYS_PER_BLOCK) void composite_kernel(
    const float* __restrict__ raw, const float* __restrict__ ts, const float* __restrict__ dirs,
    long long dirs_stride, float* __restrict__ rgb, float* __restrict__ disp,
    float* __restrict__ alpha, float* __restrict__ acc, float* __restrict__ w, long long B, int N,
    int normalize_dirs, float* __restrict__ pixels) {
    const long long ray = (long long)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= B) return;                      // whole wave leaves together; no barriers below
    const int lane = threadIdx.x & 63;
    const float* d = dirs + ray * dirs_stride;
    const float dnorm = nerf_composite::unit_dir_norm(d[0], d[1], d[2], normalize_dirs != 0);
    const GlobalSamples src{ts + ray * N, reinterpret_cast<const f32x4*>(raw) + ray * N};
    nerf_composite::composite_ray(src, N, lane, dnorm, ray, nerf_composite::RayOut{rgb, disp, alpha, acc, w, pixels});
}

// ---- backward ---------------------------------------------------------------
// d loss / d nerf_outs[B,N,4] given the upstream gradients of the five outputs
// (NULL = zero).  With G_i = dL/dw_i gathered from every consumer of w,
//   dL/dc_i     = w_i * g_rgb
//   dL/dalpha_i = G_i T_i - (1/f_i) * sum_{k>i} G_k w_k + g_alpha_i       (f = 1 - alpha + 1e-10)
//   dL/dsigma_i = dL/dalpha_i * (1 - alpha_i) * delta_i * softplus'(sigma_i)
// The suffix sum over k > i is a wave-level reverse scan; chunks of 64 samples
// are walked forward once (recomputing alpha, T, w) and backward once.
constexpr int MAX_CHUNKS = 8;          // N <= 512

__device__ __forceinline__ float wave_suffix_excl(float v, int lane, float& total) {
    // inclusive suffix sum, then shift down by one lane
    float incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float dn = __shfl_down(incl, off);
        if (lane + off < 64) incl += dn;
    }
    total = __shfl(incl, 0);
    float ex = __shfl_down(incl, 1);
    if (lane == 63) ex = 0.f;
    return ex;
}

__global__ __launch_bounds__(64 * RAYS_PER_BLOCK) void composite_backward_kernel(
    const float* __restrict__ raw, const float* __restrict__ ts, const float* __restrict__ dirs,
    long long dirs_stride, const float* __restrict__ g_rgb, const float* __restrict__ g_disp,
    const float* __restrict__ g_alpha, const float* __restrict__ g_acc, const float* __restrict__ g_w,
    float* __restrict__ d_raw, long long B, int N, int normalize_dirs,
    const float* __restrict__ mse_target, float* __restrict__ rgb_out, float mse_scale) {
    const long long ray = (long long)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= B) return;
    const int lane = threadIdx.x & 63;
    const float* d = dirs + ray * dirs_stride;
    float d0 = d[0], d1 = d[1], d2 = d[2];
    if (normalize_dirs) {
        const float n = norm3(d0, d1, d2);
        d0 = __fdiv_rn(d0, n); d1 = __fdiv_rn(d1, n); d2 = __fdiv_rn(d2, n);
    }
    const float dnorm = norm3(d0, d1, d2);
    const float* rts = ts + ray * N;
    const f32x4* rraw = reinterpret_cast<const f32x4*>(raw) + ray * N;
    f32x4* rout = reinterpret_cast<f32x4*>(d_raw) + ray * N;
    if (N == 1) {
        // the reference composites an EMPTY sample axis at N == 1 (composite_device.h): no output depends on raw
        if (lane == 0) {
            rout[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (mse_target && rgb_out) { rgb_out[ray * 3 + 0] = 0.f; rgb_out[ray * 3 + 1] = 0.f; rgb_out[ray * 3 + 2] = 0.f; }
        }
        return;
    }

    // forward sweep: per chunk keep alpha, T, fac, delta*softplus' and the colour
    float al[MAX_CHUNKS], Tt[MAX_CHUNKS], fc[MAX_CHUNKS], ds[MAX_CHUNKS], tt[MAX_CHUNKS];
    f32x4 cc[MAX_CHUNKS];
    float carry = 1.0f, depth = 0.f, accw = 0.f;
    float sr = 0.f, sg = 0.f, sb = 0.f;            // training form: the forward's rgb, for the loss gradient
#pragma unroll
    for (int ch = 0; ch < MAX_CHUNKS; ++ch) {
        const int base = ch * 64;
        al[ch] = 0.f; Tt[ch] = 0.f; fc[ch] = 1.f; ds[ch] = 0.f; tt[ch] = 0.f;
        cc[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (base < N) {
            const int i = base + lane;
            const bool valid = i < N;
            float a = 0.f, fac = 1.0f;
            if (valid) {
                const float t = rts[i];
                const f32x4 c = rraw[i];
                float delta = (i == N - 1) ? 1e10f : sub_rn(rts[i + 1], t);
                delta = mul_rn(delta, dnorm);
                const float sigma = c[3];
                const float sp = sigma > 20.f ? sigma : log1pf(expf(sigma));
                const float spd = sigma > 20.f ? 1.0f : 1.0f / (1.0f + expf(-sigma));
                a = sub_rn(1.0f, expf(mul_rn(-sp, delta)));
                fac = add_rn(sub_rn(1.0f, a), 1e-10f);
                ds[ch] = (1.0f - a) * delta * spd;       // d alpha / d sigma
                tt[ch] = t; cc[ch] = c;
            }
            // the forward compositor's scan (composite_device.h): same tree, same rounded products
            const float incl = nerf_composite::wave_scan_mul(fac);
            const float excl = nerf_composite::dpp_move<0x138, 0xf>(1.0f, incl);          // wave_shr:1
            al[ch] = a; fc[ch] = fac; Tt[ch] = mul_rn(carry, excl);
            carry = mul_rn(carry, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63)));
            if (valid) {
                depth += a * Tt[ch] * tt[ch]; accw += a * Tt[ch];
                // the same ops as the forward compositor (composite_device.h), so rgb_out equals its rgb
                const float wt = mul_rn(a, Tt[ch]);
                sr = __fmaf_rn(wt, cc[ch][0], sr); sg = __fmaf_rn(wt, cc[ch][1], sg); sb = __fmaf_rn(wt, cc[ch][2], sb);
            }
        }
    }
    depth = wave_sum(depth); accw = wave_sum(accw);

    // upstream gradients that reach every w_i of the ray
    float gr = g_rgb ? g_rgb[ray * 3 + 0] : 0.f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f,
          gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
    if (mse_target) {
        // loss = MSELoss(rgb, target) (train.py:52): d loss / d rgb = 2 (rgb - target) / (3 B), formed here
        sr = nerf_composite::wave_total(sr); sg = nerf_composite::wave_total(sg); sb = nerf_composite::wave_total(sb);
        gr = 2.0f * (sr - mse_target[ray * 3 + 0]) * mse_scale;
        gg = 2.0f * (sg - mse_target[ray * 3 + 1]) * mse_scale;
        gb = 2.0f * (sb - mse_target[ray * 3 + 2]) * mse_scale;
        if (rgb_out && lane == 0) { rgb_out[ray * 3 + 0] = sr; rgb_out[ray * 3 + 1] = sg; rgb_out[ray * 3 + 2] = sb; }
    }
    float gdep = 0.f, gac = g_acc ? g_acc[ray] : 0.f;
    if (g_disp) {
        const float q = depth / accw;
        if (q > 1e-10f) {                        // disp = 1/q there; the clamp branch has zero slope
            const float dq = -g_disp[ray] / (q * q);
            gdep = dq / accw;
            gac += -dq * depth / (accw * accw);
        }
    }
    // backward sweep over chunks, carrying sum_{k in later chunks} G_k w_k
    float later = 0.f;
#pragma unroll
    for (int ch = MAX_CHUNKS - 1; ch >= 0; --ch) {
        const int base = ch * 64;
        if (base < N) {
            const int i = base + lane;
            const bool valid = i < N;
            const float w = al[ch] * Tt[ch];
            float G = 0.f;
            if (valid) {
                G = gr * cc[ch][0] + gg * cc[ch][1] + gb * cc[ch][2] + gdep * tt[ch] + gac;
                if (g_w) G += g_w[ray * N + i];
            }
            float tot;
            const float suffix = wave_suffix_excl(valid ? G * w : 0.f, lane, tot) + later;
            later += tot;
            if (valid) {
                float dalpha = G * Tt[ch] - suffix / fc[ch];
                if (g_alpha) dalpha += g_alpha[ray * N + i];
                const f32x4 o = {w * gr, w * gg, w * gb, dalpha * ds[ch]};
                rout[i] = o;
            }
        }
    }
}

// ---- MSELoss(pred, target) and its gradient (reference train.py:52: mean over all n elements) ----
// One workgroup, fixed summation order: the loss is bit-reproducible from run to run.
__global__ __launch_bounds__(1024) void mse_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        float* __restrict__ loss, float* __restrict__ g_pred, long long n) {
    __shared__ float red[16];
    const float inv_n = 1.0f / (float)n;
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const float d = pred[i] - target[i];
        s += d * d;
        if (g_pred) g_pred[i] = 2.0f * d * inv_n;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += red[k];
        *loss = t * inv_n;
    }
}

}  // namespace

extern "C" int nerf_amd_launch_composite_backward(const float* raw, const float* ts, const float* dirs,
                                                  long long dirs_stride, const float* g_rgb,
                                                  const float* g_disp, const float* g_alpha,
                                                  const float* g_acc, const float* g_w, float* d_raw,
                                                  long long B, int N, int normalize_dirs, hipStream_t stream) {
    (void)hipGetLastError();
    if (B == 0) return 0;
    if (N > 64 * MAX_CHUNKS) return -2;
    const long long blocks = (B + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK;
    hipLaunchKernelGGL(composite_backward_kernel, dim3((unsigned)blocks), dim3(64 * RAYS_PER_BLOCK), 0, stream,
                       raw, ts, dirs, dirs_stride, g_rgb, g_disp, g_alpha, g_acc, g_w, d_raw, B, N,
                       normalize_dirs, nullptr, nullptr, 0.f);
    return (int)hipGetLastError();
}

// training form: compositing forward + MSELoss gradient + compositing backward in one launch
extern "C" int nerf_amd_launch_composite_mse_backward(const float* raw, const float* ts, const float* rays,
                                                      const float* target, float* rgb, float* d_raw, long long B,
                                                      int N, hipStream_t stream) {
    (void)hipGetLastError();
    if (B == 0) return 0;
    if (N > 64 * MAX_CHUNKS) return -2;
    const long long blocks = (B + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK;
    hipLaunchKernelGGL(composite_backward_kernel, dim3((unsigned)blocks), dim3(64 * RAYS_PER_BLOCK), 0, stream,
                       raw, ts, rays + 3, 6ll, nullptr, nullptr, nullptr, nullptr, nullptr, d_raw, B, N, 1,
                       target, rgb, 1.0f / (3.0f * (float)B));
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_composite(const float* raw, const float* ts, const float* dirs,
                                         long long dirs_stride, float* rgb, float* disp, float* alpha,
                                         float* acc, float* w, long long B, int N, int normalize_dirs,
                                         float* pixels, hipStream_t stream) {
    (void)hipGetLastError();   // drop any stale error: the return value is about THIS launch
    if (B == 0) return 0;
    const long long blocks = (B + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)blocks), dim3(64 * RAYS_PER_BLOCK), 0, stream,
                       raw, ts, dirs, dirs_stride, rgb, disp, alpha, acc, w, B, N, normalize_dirs, pixels);
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_mse_loss(const float* pred, const float* target, float* loss, float* g_pred, long long n,
                                        hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(1024), 0, stream, pred, target, loss, g_pred, n);
    return (int)hipGetLastError();
}
