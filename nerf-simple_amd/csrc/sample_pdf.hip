// sample_pdf.hip -- hierarchical ("fine") sample placement for BASELINE config 4
// (64 coarse + 128 importance samples).  ABSENT from the reference
// (README.md:3, configs/lego.yaml:7, utils/nets.py:45-49): parity UNPINNED; this
// follows the NeRF paper's sample_pdf (inverse-CDF sampling of the coarse
// weights' interior bins) and is checked against oracle/nerf_oracle.sample_pdf.
//
//   mids  = (ts[1:] + ts[:-1]) / 2                      Nc-1 bin edges
//   pdf   = (w[1:-1] + 1e-5) / sum                      Nc-2 bins
//   cdf   = [0, cumsum(pdf)]                            Nc-1 values
//   z     = inverse-cdf(u), linear inside a bin         Nf new positions
//   out   = sort(concat(ts, z))                         Nc+Nf positions per ray
//
// One wavefront per ray; everything lives in that wave's slice of LDS: the cdf
// is a wave-level inclusive sum scan, each lane inverts the cdf for its own u by
// binary search, and the merged list is sorted by a bitonic network over 512
// LDS slots (padded with +inf).  Nc <= 256, Nc + Nf <= 512.
#include "nerf_device.h"

namespace {

constexpr int RPB = 4;            // rays (waves) per block
constexpr int MAXC = 256;
constexpr int MAXM = 512;

__global__ __launch_bounds__(64 * RPB) void sample_pdf_kernel(
    const float* __restrict__ ts, const float* __restrict__ w, const float* __restrict__ u,
    float* __restrict__ out, long long B, int Nc, int Nf, unsigned long long seed, long long ray_id0,
    int device_rng) {
    __shared__ float s_cdf[RPB][MAXC];
    __shared__ float s_bins[RPB][MAXC];
    __shared__ float s_all[RPB][MAXM];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * RPB + wv;
    if (ray >= B) return;                              // wave-uniform; only wave-local LDS is used
    float* cdf = s_cdf[wv];
    float* bins = s_bins[wv];
    float* all = s_all[wv];
    const float* rts = ts + ray * Nc;
    const float* rw = w + ray * Nc;
    const int nb = Nc - 1;                             // bin edges (mids); nb-1 bins

    // mids, coarse positions into the merge buffer, bin weights
    float local = 0.f;
    for (int i = lane; i < MAXM; i += 64) all[i] = i < Nc ? rts[i] : __builtin_inff();
    for (int i = lane; i < nb; i += 64) bins[i] = 0.5f * (rts[i + 1] + rts[i]);
    // inclusive scan of (w[1:-1] + 1e-5) in chunks of 64, cdf[0] = 0
    float carry = 0.f;
    for (int base = 0; base < nb - 1; base += 64) {
        const int i = base + lane;
        float v = i < nb - 1 ? rw[i + 1] + 1e-5f : 0.f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float up = __shfl_up(v, off);
            if (lane >= off) v += up;
        }
        if (i < nb - 1) cdf[i + 1] = carry + v;
        carry += __shfl(v, 63);
    }
    if (lane == 0) cdf[0] = 0.f;
    const float total = carry;
    (void)local;
    __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): this wave's LDS writes are done
    __builtin_amdgcn_wave_barrier();

    // inverse cdf for each new sample
    for (int j = lane; j < Nf; j += 64) {
        float uu;
        if (device_rng) uu = philox_uniform(seed ^ 0x9e3779b97f4a7c15ull, (unsigned long long)((ray_id0 + ray) * Nf + j));
        else uu = u[ray * Nf + j];
        const float target = uu * total;               // cdf kept un-normalised: compare against u * sum
        // searchsorted(cdf, target, side='right') over cdf[0..nb-1]
        int lo = 0, hi = nb;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= target) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nb - 1 ? lo : nb - 1;
        const float c0 = cdf[below] / total, c1 = cdf[above] / total;
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.f;
        const float tt = (uu - c0) / denom;
        all[Nc + j] = bins[below] + tt * (bins[above] - bins[below]);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    // bitonic sort of 512 slots by one wave (8 slots per lane)
    for (int k = 2; k <= MAXM; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < MAXM; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const float a = all[i], b = all[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { all[i] = b; all[l] = a; }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
    const int M = Nc + Nf;
    for (int i = lane; i < M; i += 64) out[ray * M + i] = all[i];
}

}  // namespace

extern "C" int nerf_amd_launch_sample_pdf(const float* ts, const float* w, const float* u, float* out,
                                          long long B, int Nc, int Nf, unsigned long long seed,
                                          long long ray_id0, int device_rng, hipStream_t stream) {
    (void)hipGetLastError();
    if (B == 0) return 0;
    if (Nc < 3 || Nc > MAXC || Nf < 0 || Nc + Nf > MAXM) return -2;
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)((B + RPB - 1) / RPB)), dim3(64 * RPB), 0, stream,
                       ts, w, u, out, B, Nc, Nf, seed, ray_id0, device_rng);
    return (int)hipGetLastError();
}
