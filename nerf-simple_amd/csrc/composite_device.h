// composite_device.h -- the per-ray compositing routine (reference utils/rendering.py:47-85), shared
// by composite.hip (samples read from HBM) and the fused render kernel in mlp_bf16_16.hip (samples
// read from the workgroup's LDS ring).  ONE wavefront per ray, one sample per lane, N walked in
// chunks of 64 with the transmittance carried between chunks.  Every floating-point operation is
// written as one explicitly rounded op (no contraction left to the compiler), so both callers
// produce bit-identical results from bit-identical samples.
#pragma once
#include "nerf_device.h"

namespace nerf_composite {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = add_rn(v, __shfl_xor(v, off));
    return v;
}

// Cross-lane moves as DPP modifiers (VALU speed) instead of ds_bpermute round trips through the LDS
// crossbar: row_shr:n inside each row of 16 lanes, row_bcast:15 / row_bcast:31 between rows.  A lane
// with no source (or in a row ROW_MASK leaves out) receives `idle`, the identity of the caller's op.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float idle, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(idle), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// inclusive product scan over the 64 lanes; every product is one rounded multiply
__device__ __forceinline__ float wave_scan_mul(float x) {
    x = mul_rn(x, dpp_move<0x111, 0xf>(1.0f, x));      // row_shr:1
    x = mul_rn(x, dpp_move<0x112, 0xf>(1.0f, x));      // row_shr:2
    x = mul_rn(x, dpp_move<0x114, 0xf>(1.0f, x));      // row_shr:4
    x = mul_rn(x, dpp_move<0x118, 0xf>(1.0f, x));      // row_shr:8
    x = mul_rn(x, dpp_move<0x142, 0xa>(1.0f, x));      // row_bcast:15 -> rows 1, 3
    x = mul_rn(x, dpp_move<0x143, 0xc>(1.0f, x));      // row_bcast:31 -> rows 2, 3
    return x;
}
// sum over the 64 lanes as a wave-uniform value (the same tree with adds; lane 63 holds the total)
__device__ __forceinline__ float wave_total(float x) {
    x = add_rn(x, dpp_move<0x111, 0xf>(0.f, x));
    x = add_rn(x, dpp_move<0x112, 0xf>(0.f, x));
    x = add_rn(x, dpp_move<0x114, 0xf>(0.f, x));
    x = add_rn(x, dpp_move<0x118, 0xf>(0.f, x));
    x = add_rn(x, dpp_move<0x142, 0xa>(0.f, x));
    x = add_rn(x, dpp_move<0x143, 0xc>(0.f, x));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// ||d / ||d|| || as the reference computes it: dirs = rays[:,3:] / norm (rendering.py:37), then
// torch.norm(dirs[..., None, :], dim=-1) inside volume_render (:62)
__device__ __forceinline__ float unit_dir_norm(float d0, float d1, float d2, bool normalize) {
    if (normalize) {
        const float n = norm3(d0, d1, d2);
        d0 = __fdiv_rn(d0, n); d1 = __fdiv_rn(d1, n); d2 = __fdiv_rn(d2, n);
    }
    return norm3(d0, d1, d2);
}

struct RayOut {            // any pointer may be NULL
    float* rgb;            // [B,3]
    float* disp;           // [B]
    float* alpha;          // [B,N]
    float* acc;            // [B]
    float* w;              // [B,N]
    float* pixels;         // [B,4] = [clip(rgb,0,1), disparity]  (image drivers, rendering.py:103-105)
};

// Src: float t(int i) const; f32x4 c(int i) const;  -- sample i of THIS ray (0 <= i < N)
template <class Src>
__device__ __forceinline__ void composite_ray(const Src& src, int N, int lane, float dnorm, long long ray,
                                              const RayOut& o) {
    float carry = 1.0f;                        // transmittance entering this chunk
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
    // N == 1 in the reference: deltas = cat(ts[:,1:] - ts[:,:-1], 1e10 * ones_like(deltas[:, :1])) is built from an
    // EMPTY [B,0] difference, so the sample axis stays empty (utils/rendering.py:60-61): alpha / w are [B,0],
    // rgb = acc = 0 and disparity = 1 / max(1e-10, 0/0) = NaN.  Reproduced by compositing no sample at all.
    if (N == 1) N = 0;
    for (int base = 0; base < N; base += 64) {
        const int i = base + lane;
        const bool valid = i < N;
        float a = 0.f, t = 0.f, fac = 1.0f;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            t = src.t(i);
            c = src.c(i);
            float delta = (i == N - 1) ? 1e10f : sub_rn(src.t(i + 1), t);
            delta = mul_rn(delta, dnorm);
            const float sigma = c[3];
            const float sp = sigma > 20.f ? sigma : log1pf(expf(sigma));      // softplus(beta=1, threshold=20)
            a = sub_rn(1.0f, expf(mul_rn(-sp, delta)));
            fac = add_rn(sub_rn(1.0f, a), 1e-10f);
        }
        // inclusive product scan across the wave, shifted by one lane = exclusive cumprod (rendering.py:68)
        const float incl = wave_scan_mul(fac);
        const float excl = dpp_move<0x138, 0xf>(1.0f, incl);          // wave_shr:1
        const float T = mul_rn(carry, excl);
        const float wt = mul_rn(a, T);
        carry = mul_rn(carry, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63)));
        if (valid) {
            // per-sample outputs are streamed (8 B per sample when requested): non-temporal stores
            if (o.alpha) __builtin_nontemporal_store(a, o.alpha + ray * N + i);
            if (o.w) __builtin_nontemporal_store(wt, o.w + ray * N + i);
            sr = __fmaf_rn(wt, c[0], sr);
            sg = __fmaf_rn(wt, c[1], sg);
            sb = __fmaf_rn(wt, c[2], sb);
            sd = __fmaf_rn(wt, t, sd);
            sa = add_rn(sa, wt);
        }
    }
    sr = wave_total(sr); sg = wave_total(sg); sb = wave_total(sb);
    sd = wave_total(sd); sa = wave_total(sa);
    if (lane == 0) {
        const float q = __fdiv_rn(sd, sa);
        const float m = (q != q) ? q : fmaxf(1e-10f, q);   // torch.max propagates NaN
        const float dsp = __fdiv_rn(1.0f, m);
        if (o.rgb) { o.rgb[ray * 3 + 0] = sr; o.rgb[ray * 3 + 1] = sg; o.rgb[ray * 3 + 2] = sb; }
        if (o.acc) o.acc[ray] = sa;
        if (o.disp) o.disp[ray] = dsp;
        if (o.pixels) {
            // clip rgb to [0,1] AFTER compositing (torch.clip passes NaN through), disparity un-clipped
            auto clip01 = [](float v) { return (v != v) ? v : fminf(fmaxf(v, 0.f), 1.f); };
            const f32x4 px = {clip01(sr), clip01(sg), clip01(sb), dsp};
            *reinterpret_cast<f32x4*>(o.pixels + ray * 4) = px;
        }
    }
}

}  // namespace nerf_composite
