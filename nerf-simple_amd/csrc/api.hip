// api.hip -- the C ABI of libnerf_amd.so (include/nerf_amd.h): argument
// checking, host-side introspection and the composition of the render path
// out of the kernels in this directory.  No allocation, no host sync.
#include "nerf_device.h"
#include "../../include/nerf_amd.h"
#include <math.h>

using namespace nerf_layout;

extern "C" {
int nerf_amd_launch_pack(const float*, void*, int, hipStream_t);
int nerf_amd_launch_pack_train(const float*, void*, void*, hipStream_t);
int nerf_amd_launch_composite_mse_backward(const float*, const float*, const float*, const float*, float*, float*, long long,
                                           int, hipStream_t);
int nerf_amd_launch_param_gradients_begin(const float*, void*, float*, long long, hipStream_t);
int nerf_amd_launch_param_gradients_finish(const void*, const void*, const void*, const void*, const void*, float*, long long,
                                           int, hipStream_t);
int nerf_amd_launch_query_points(const MlpArgs*, float*, hipStream_t);
int nerf_amd_launch_gamma(const float*, long long, float*, long long, int, hipStream_t);
int nerf_amd_launch_posenc(const float*, float*, float*, long long, int, int, hipStream_t);
int nerf_amd_launch_composite(const float*, const float*, const float*, long long, float*, float*,
                              float*, float*, float*, long long, int, int, float*, hipStream_t);
int nerf_amd_launch_sample_pdf(const float*, const float*, const float*, float*, long long, int, int,
                               unsigned long long, long long, int, hipStream_t);
int nerf_amd_launch_generate_rays(const float*, int, int, float, long long, long long, float*, hipStream_t);
int nerf_amd_launch_composite_backward(const float*, const float*, const float*, long long, const float*,
                                       const float*, const float*, const float*, const float*, float*,
                                       long long, int, int, hipStream_t);
int nerf_amd_launch_mse_loss(const float*, const float*, float*, float*, long long, hipStream_t);
int nerf_amd_launch_sample_encode(const MlpArgs*, float*, float*, hipStream_t);
int nerf_amd_launch_mlp_f32(const MlpArgs*, int, hipStream_t);
int nerf_amd_launch_mlp_bf16_16(const MlpArgs*, int, hipStream_t);
int nerf_amd_launch_mlp_f16_16(const MlpArgs*, int, hipStream_t);
int nerf_amd_launch_mlp_backward(const float*, const void*, const void*, void*, long long, int, hipStream_t);
int nerf_amd_launch_param_gradients_finish_e4m3(const void*, const void*, const void*, float*, long long, int, hipStream_t);
int nerf_amd_launch_param_gradients_convert_e4m3(const void*, const void*, const void*, void*, long long, int, hipStream_t);
long long nerf_amd_f8_scratch_bytes(long long);
int nerf_amd_launch_mt19937_uniform(const uint32_t*, int, float*, long long, uint32_t*, hipStream_t);
int nerf_amd_launch_mt19937_uniform_par(const uint32_t*, int, float*, long long, uint32_t*, const uint32_t*, int, long long,
                                        uint32_t*, hipStream_t);
int nerf_amd_launch_range_check(const MlpArgs*, long long, unsigned*, hipStream_t);
int nerf_amd_launch_mt19937_raw(const uint32_t*, int, uint32_t*, long long, uint32_t*, hipStream_t);
int nerf_amd_launch_mt19937_uniform_after(const uint32_t*, const uint32_t*, int, int, float*, long long, uint32_t*, long long, uint32_t*,
                                          hipStream_t);
int nerf_amd_launch_mt19937_advance(const uint32_t*, const uint32_t*, uint32_t*, hipStream_t);
int nerf_amd_launch_select_rays(const uint32_t*, unsigned long long, const unsigned long long*, long long, long long, const float*,
                                const float*, float*, float*, long long*, void*, hipStream_t);
int nerf_amd_host_mt19937_jump_poly(long long, const uint32_t*, uint32_t*);
int nerf_amd_launch_adam_hyper(float*, const float*, float*, float*, long long, const float*, hipStream_t);
int nerf_amd_launch_hyper_fetch(const float*, int, float*, unsigned*, hipStream_t);
int nerf_amd_launch_linear_f32(const float*, long long, long long, const float*, const float*, long long, long long, const float*,
                               float*, long long, long long, long long, long long, int, hipStream_t);
int nerf_amd_launch_adam(float*, const float*, float*, float*, long long, float, float, float, float, float, float,
                         hipStream_t);
int nerf_amd_launch_sample_encode_bf16(const MlpArgs*, void*, void*, hipStream_t);
int nerf_amd_launch_param_gradients(const float*, const void*, const void*, const void*, const void*, void*, float*,
                                    long long, hipStream_t);
}

namespace {
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline bool bad_precision(int p) { return p != NERF_AMD_F32 && p != NERF_AMD_BF16 && p != NERF_AMD_FP16; }
inline bool bad_image(int p) { return bad_precision(p) && p != NERF_AMD_BF16_BWD; }
inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
// the jitter arguments of every rays-mode entry point: explicit u / ts, the counter RNG, or the counter RNG with its
// seed offset in device memory (then `u` is that address)
inline bool bad_jitter(uint32_t flags, const float* u, const float* tbins) {
    if (flags & ~(NERF_AMD_TS_GIVEN | NERF_AMD_DEVICE_RNG | NERF_AMD_SEED_IN_MEMORY)) return true;     // unknown bits
    if (flags & NERF_AMD_SEED_IN_MEMORY) {
        if (!(flags & NERF_AMD_DEVICE_RNG) || (flags & NERF_AMD_TS_GIVEN) || !u) return true;
        if (reinterpret_cast<uintptr_t>(u) & 7) return true;                   // the kernels load it as one 64-bit word
    } else if (!(flags & NERF_AMD_DEVICE_RNG) && !u) {
        return true;
    }
    return !(flags & NERF_AMD_TS_GIVEN) && !tbins;
}

// the fused render kernels (sampling + MLP + compositing in one launch) serve rays of up to
// FUSED_RENDER_MAX_N samples, in every precision
inline bool fused_render(int precision, int N) { return !bad_precision(precision) && N <= FUSED_RENDER_MAX_N; }
int launch_mlp(const MlpArgs& a, int rays_mode, int precision, hipStream_t s) {
    if (precision == NERF_AMD_F32) return nerf_amd_launch_mlp_f32(&a, rays_mode, s);
    if (precision == NERF_AMD_FP16) return nerf_amd_launch_mlp_f16_16(&a, rays_mode, s);
    return nerf_amd_launch_mlp_bf16_16(&a, rays_mode, s);
}
}  // namespace

extern "C" {

int nerf_amd_abi_version(void) { return NERF_AMD_ABI_VERSION; }
int64_t nerf_amd_param_count(void) { return PARAM_COUNT; }

int64_t nerf_amd_packed_bytes(int precision) {
    if (bad_image(precision)) return NERF_AMD_EINVAL;
    if (precision == NERF_AMD_BF16_BWD) return BWD_IMAGE_BYTES;
    return precision == NERF_AMD_F32 ? F32_PACKED_BYTES : B16_IMAGE_BYTES;
}

int64_t nerf_amd_packed_status_offset(int precision) {
    if (bad_image(precision)) return NERF_AMD_EINVAL;
    return (precision == NERF_AMD_BF16 || precision == NERF_AMD_FP16) ? B16_STATUS_OFF : -1;
}

int nerf_amd_grad_bucket_range(int bucket, int64_t* first, int64_t* count) {
    if (!first || !count || bucket < 0 || bucket > 2) return NERF_AMD_EINVAL;
    *first = bucket == 1 ? GRAD_BUCKET_SPLIT : 0;
    *count = bucket == 0 ? PARAM_COUNT : bucket == 1 ? PARAM_COUNT - GRAD_BUCKET_SPLIT : GRAD_BUCKET_SPLIT;
    return 0;
}

int64_t nerf_amd_render_image_workspace_bytes(int precision, int64_t n_rays, int N) {
    if (n_rays < 0 || N <= 0 || bad_precision(precision)) return NERF_AMD_EINVAL;
    // rays[n,6] (+ raw[n,N,4] + ts[n,N] on the two-launch path)
    if (fused_render(precision, N)) return align_up(n_rays * 24, 256);
    return align_up(n_rays * 24, 256) + align_up(n_rays * N * 16, 256) + align_up(n_rays * N * 4, 256);
}

int64_t nerf_amd_render_workspace_bytes(int precision, int64_t B, int N) {
    if (B < 0 || N <= 0 || bad_precision(precision)) return NERF_AMD_EINVAL;
    if (fused_render(precision, N)) return 0;
    // raw[B,N,4] + ts[B,N]
    return align_up(B * N * 16, 256) + align_up(B * N * 4, 256);
}

int nerf_amd_layout_src_col(int precision, int layer, int kstep, int half, int elem) {
    if (layer < 0 || layer >= NUM_LAYERS) return -2;
    if (precision == NERF_AMD_BF16 || precision == NERF_AMD_FP16) {
        if (kstep < 0 || kstep >= b16_ks(layer) || half < 0 || half > 3 || elem < 0 || elem > 7) return -2;
        return src_col_b16(layer, kstep, half, elem);
    }
    if (precision == NERF_AMD_F32) {
        if (kstep < 0 || kstep >= f32_ks(layer) || half < 0 || half > 3) return -2;
        return src_col_f32(layer, kstep, half);
    }
    return -2;
}

// Every source column of every layer must be hit exactly once by the packed
// k positions (the permutations are bijections onto the true K, padding aside).
int nerf_amd_layout_selfcheck(void) {
    static_assert(F32_NUM_CHUNKS == 154, "f32 chunk count");
    static_assert(B16_WEIGHT_KIB == 1172 && B16_BIAS_FLOATS == F32_BIAS_FLOATS, "16-row 16-bit image");
    for (int L = 0; L < NUM_LAYERS; ++L) {
        const LayerDesc d = layer_desc(L);
        for (int prec = 0; prec < 2; ++prec) {          // 0: f32 k order, 1: 16-bit k order
            int seen[320] = {0}, pads = 0;
            if (prec == 1) {
                for (int s = 0; s < b16_ks(L); ++s)
                    for (int g = 0; g < 4; ++g)
                        for (int j = 0; j < 8; ++j) {
                            const int c = src_col_b16(L, s, g, j);
                            if (c < 0) { ++pads; continue; }
                            if (c >= d.ld) return 100 + L;
                            ++seen[c];
                        }
            } else {
                for (int s = 0; s < f32_ks(L); ++s)
                    for (int g = 0; g < 4; ++g) {
                        const int c = src_col_f32(L, s, g);
                        if (c < 0) { ++pads; continue; }
                        if (c >= d.ld) return 200 + L;
                        ++seen[c];
                    }
            }
            for (int i = 0; i < d.ld; ++i)
                if (seen[i] != 1) return 300 + 20 * prec + L;
            if (pads != layer_k(L) - d.ld) return 400 + 20 * prec + L;
        }
    }
    return 0;
}

int nerf_amd_pack_weights(const float* params, void* packed, int precision, void* stream) {
    if (!params || !packed || bad_image(precision)) return NERF_AMD_EINVAL;
    return nerf_amd_launch_pack(params, packed, precision, S(stream));
}

int nerf_amd_pack_weights_train(const float* params, void* packed_bf16, void* packed_bwd, void* stream) {
    if (!params || !packed_bf16 || !packed_bwd) return NERF_AMD_EINVAL;
    return nerf_amd_launch_pack_train(params, packed_bf16, packed_bwd, S(stream));
}

int nerf_amd_gamma(const float* x, int64_t x_stride, float* out, int64_t n, int L, void* stream) {
    if (n < 0 || L < 0 || x_stride < 0) return NERF_AMD_EINVAL;
    if (n == 0 || L == 0) return 0;
    if (!x || !out) return NERF_AMD_EINVAL;
    return nerf_amd_launch_gamma(x, x_stride, out, n, L, S(stream));
}

int nerf_amd_positional_encoder(const float* vec, float* posx, float* posd, int64_t P, int Lp, int Ld,
                                void* stream) {
    if (P < 0 || Lp < 0 || Ld < 0) return NERF_AMD_EINVAL;
    if (P == 0) return 0;
    if (!vec || !posx || !posd) return NERF_AMD_EINVAL;
    return nerf_amd_launch_posenc(vec, posx, posd, P, Lp, Ld, S(stream));
}

int nerf_amd_query_points(const float* rays, const float* u, const float* tbins, uint32_t flags, uint64_t seed,
                          int64_t ray_id0, float* query_pts, float* ts, int64_t B, int N, void* stream) {
    if (B < 0 || N <= 0) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (!rays || !query_pts) return NERF_AMD_EINVAL;
    if (bad_jitter(flags, u, tbins)) return NERF_AMD_EINVAL;
    MlpArgs a{};
    a.rays = rays; a.u = u; a.tbins = tbins; a.ts_out = ts;
    a.P = B * (int64_t)N; a.N = N; a.flags = flags; a.seed = seed; a.ray_id0 = ray_id0;
    return nerf_amd_launch_query_points(&a, query_pts, S(stream));
}

int nerf_amd_range_check(const float* rays, const float* pts, const float* u, const float* tbins, uint32_t flags, uint64_t seed,
                         int64_t ray_id0, uint32_t* word, int64_t B, int N, void* stream) {
    if (B < 0 || !word || (rays && pts)) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    MlpArgs a{};
    if (pts) {
        a.pts = pts; a.P = B;          // B floats
    } else {
        if (!rays || N <= 0) return NERF_AMD_EINVAL;
        if (bad_jitter(flags, u, tbins)) return NERF_AMD_EINVAL;
        a.rays = rays; a.u = u; a.tbins = tbins;
        a.P = B * (int64_t)N; a.N = N; a.flags = flags; a.seed = seed; a.ray_id0 = ray_id0;
    }
    return nerf_amd_launch_range_check(&a, B, word, S(stream));
}

int nerf_amd_mlp_forward(const float* pts, void* packed, float* out, int64_t P, int precision,
                         void* stream) {
    if (P < 0 || bad_precision(precision)) return NERF_AMD_EINVAL;
    if (P == 0) return 0;
    if (!pts || !packed || !out) return NERF_AMD_EINVAL;
    MlpArgs a{};
    a.pts = pts; a.packed = packed; a.raw = out; a.P = P; a.N = 1;
    return launch_mlp(a, 0, precision, S(stream));
}

int nerf_amd_volume_render(const float* raw, const float* ts, const float* dirs, int64_t dirs_stride,
                           float* rgb, float* disp, float* alpha, float* acc, float* w, int64_t B, int N,
                           void* stream) {
    if (B < 0 || N <= 0 || dirs_stride < 3) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (!raw || !ts || !dirs || !rgb || !disp || !acc) return NERF_AMD_EINVAL;
    return nerf_amd_launch_composite(raw, ts, dirs, dirs_stride, rgb, disp, alpha, acc, w, B, N, 0, nullptr, S(stream));
}

int nerf_amd_volume_render_pixels(const float* raw, const float* ts, const float* rays, float* pixels,
                                  int64_t B, int N, void* stream) {
    if (B < 0 || N <= 0) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (!raw || !ts || !rays || !pixels) return NERF_AMD_EINVAL;
    return nerf_amd_launch_composite(raw, ts, rays + 3, 6, nullptr, nullptr, nullptr, nullptr, nullptr, B, N, 1,
                                     pixels, S(stream));
}

int nerf_amd_volume_render_rays(const float* raw, const float* ts, const float* rays, float* rgb, float* disp,
                                float* alpha, float* acc, float* w, int64_t B, int N, void* stream) {
    if (B < 0 || N <= 0) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (!raw || !ts || !rays || !rgb || !disp || !acc) return NERF_AMD_EINVAL;
    return nerf_amd_launch_composite(raw, ts, rays + 3, 6, rgb, disp, alpha, acc, w, B, N, 1, nullptr, S(stream));
}

int nerf_amd_volume_render_rays_backward(const float* raw, const float* ts, const float* rays, const float* g_rgb,
                                         const float* g_disp, const float* g_alpha, const float* g_acc,
                                         const float* g_w, float* d_raw, int64_t B, int N, void* stream) {
    if (B < 0 || N <= 0) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (N > 512) return NERF_AMD_EUNSUP;
    if (!raw || !ts || !rays || !d_raw) return NERF_AMD_EINVAL;
    return nerf_amd_launch_composite_backward(raw, ts, rays + 3, 6, g_rgb, g_disp, g_alpha, g_acc, g_w, d_raw, B, N, 1,
                                              S(stream));
}

int nerf_amd_volume_render_mse_backward(const float* raw, const float* ts, const float* rays, const float* target,
                                       float* rgb, float* d_raw, int64_t B, int N, void* stream) {
    if (B < 0 || N <= 0) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (N > 512) return NERF_AMD_EUNSUP;
    if (!raw || !ts || !rays || !target || !d_raw) return NERF_AMD_EINVAL;
    return nerf_amd_launch_composite_mse_backward(raw, ts, rays, target, rgb, d_raw, B, N, S(stream));
}

int nerf_amd_mse_loss(const float* pred, const float* target, float* loss, float* g_pred, int64_t n, void* stream) {
    if (n <= 0 || !pred || !target || !loss) return NERF_AMD_EINVAL;
    return nerf_amd_launch_mse_loss(pred, target, loss, g_pred, n, S(stream));
}

int nerf_amd_volume_render_backward(const float* raw, const float* ts, const float* dirs, int64_t dirs_stride,
                                    const float* g_rgb, const float* g_disp, const float* g_alpha,
                                    const float* g_acc, const float* g_w, float* d_raw, int64_t B, int N,
                                    void* stream) {
    if (B < 0 || N <= 0 || dirs_stride < 3) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (N > 512) return NERF_AMD_EUNSUP;
    if (!raw || !ts || !dirs || !d_raw) return NERF_AMD_EINVAL;
    return nerf_amd_launch_composite_backward(raw, ts, dirs, dirs_stride, g_rgb, g_disp, g_alpha, g_acc, g_w,
                                              d_raw, B, N, 0, S(stream));
}

int nerf_amd_sample_encode(const float* rays, const float* u, const float* tbins, uint32_t flags,
                           uint64_t seed, int64_t ray_id0, float* posx, float* posd, float* ts, int64_t B,
                           int N, void* stream) {
    if (B < 0 || N <= 0) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (!rays || !posx || !posd) return NERF_AMD_EINVAL;
    if (bad_jitter(flags, u, tbins)) return NERF_AMD_EINVAL;
    MlpArgs a{};
    a.rays = rays; a.u = u; a.tbins = tbins; a.ts_out = ts;
    a.P = B * (int64_t)N; a.N = N; a.flags = flags; a.seed = seed; a.ray_id0 = ray_id0;
    return nerf_amd_launch_sample_encode(&a, posx, posd, S(stream));
}

int nerf_amd_mlp_forward_rays(const float* rays, const float* u, const float* tbins, void* packed,
                              int precision, uint32_t flags, uint64_t seed, int64_t ray_id0, float* raw,
                              float* ts, int64_t B, int N, void* stream) {
    if (B < 0 || N <= 0 || bad_precision(precision)) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (!rays || !packed || !raw) return NERF_AMD_EINVAL;
    if (bad_jitter(flags, u, tbins)) return NERF_AMD_EINVAL;
    MlpArgs a{};
    a.rays = rays; a.u = u; a.tbins = tbins; a.packed = packed; a.raw = raw; a.ts_out = ts;
    a.P = B * (int64_t)N; a.N = N; a.flags = flags; a.seed = seed; a.ray_id0 = ray_id0;
    return launch_mlp(a, 1, precision, S(stream));
}

int nerf_amd_render_forward(const float* rays, const float* u, const float* tbins, void* packed,
                            int precision, uint32_t flags, uint64_t seed, int64_t ray_id0, float* rgb,
                            float* disp, float* alpha, float* acc, float* w, void* workspace, int64_t B,
                            int N, void* stream) {
    if (B < 0 || N <= 0 || bad_precision(precision)) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (!rgb || !disp || !acc) return NERF_AMD_EINVAL;
    if (fused_render(precision, N)) {
        if (!rays || !packed) return NERF_AMD_EINVAL;
        if (bad_jitter(flags, u, tbins)) return NERF_AMD_EINVAL;
        MlpArgs a{};
        a.rays = rays; a.u = u; a.tbins = tbins; a.packed = packed;
        a.P = B * (int64_t)N; a.N = N; a.flags = flags; a.seed = seed; a.ray_id0 = ray_id0;
        a.rgb = rgb; a.disp = disp; a.alpha = alpha; a.acc = acc; a.w = w;
        return launch_mlp(a, 1, precision, S(stream));
    }
    if (!workspace) return NERF_AMD_EINVAL;
    float* raw = reinterpret_cast<float*>(workspace);
    float* ts = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + align_up(B * N * 16, 256));
    int rc = nerf_amd_mlp_forward_rays(rays, u, tbins, packed, precision, flags, seed, ray_id0, raw, ts, B, N,
                                       stream);
    if (rc) return rc;
    // dirs = rays[:,3:] normalised inside the kernel (utils/rendering.py:37,43)
    return nerf_amd_launch_composite(raw, ts, rays + 3, 6, rgb, disp, alpha, acc, w, B, N, 1, nullptr, S(stream));
}

int nerf_amd_render_pixels_forward(const float* rays, const float* u, const float* tbins, void* packed,
                                   int precision, uint32_t flags, uint64_t seed, int64_t ray_id0, float* pixels,
                                   void* workspace, int64_t B, int N, void* stream) {
    if (B < 0 || N <= 0 || bad_precision(precision)) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (!pixels) return NERF_AMD_EINVAL;
    if (fused_render(precision, N)) {
        if (!rays || !packed) return NERF_AMD_EINVAL;
        if (bad_jitter(flags, u, tbins)) return NERF_AMD_EINVAL;
        MlpArgs a{};
        a.rays = rays; a.u = u; a.tbins = tbins; a.packed = packed;
        a.P = B * (int64_t)N; a.N = N; a.flags = flags; a.seed = seed; a.ray_id0 = ray_id0;
        a.pixels = pixels;
        return launch_mlp(a, 1, precision, S(stream));
    }
    if (!workspace) return NERF_AMD_EINVAL;
    float* raw = reinterpret_cast<float*>(workspace);
    float* ts = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + align_up(B * N * 16, 256));
    int rc = nerf_amd_mlp_forward_rays(rays, u, tbins, packed, precision, flags, seed, ray_id0, raw, ts, B, N,
                                       stream);
    if (rc) return rc;
    return nerf_amd_launch_composite(raw, ts, rays + 3, 6, nullptr, nullptr, nullptr, nullptr, nullptr, B, N, 1,
                                     pixels, S(stream));
}

int nerf_amd_generate_rays(const float* h_pose, int H, int W, float f, int64_t ray0, int64_t n_rays,
                           float* rays, void* stream) {
    if (!h_pose || H <= 0 || W <= 0 || !(f > 0.f) || ray0 < 0 || n_rays < 0 ||
        ray0 + n_rays > (int64_t)H * W) return NERF_AMD_EINVAL;
    if (n_rays == 0) return 0;
    if (!rays) return NERF_AMD_EINVAL;
    return nerf_amd_launch_generate_rays(h_pose, H, W, f, ray0, n_rays, rays, S(stream));
}

int nerf_amd_render_image_forward(const float* h_pose, int H, int W, float f, int64_t ray0, int64_t n_rays,
                                  const float* u, const float* tbins, void* packed, int precision,
                                  uint32_t flags, uint64_t seed, float* pixels, void* workspace, int N,
                                  void* stream) {
    if (n_rays < 0 || N <= 0 || bad_precision(precision)) return NERF_AMD_EINVAL;
    if (n_rays == 0) return 0;
    if (!workspace || !pixels || !packed) return NERF_AMD_EINVAL;
    char* ws = reinterpret_cast<char*>(workspace);
    float* rays = reinterpret_cast<float*>(ws);
    int rc = nerf_amd_generate_rays(h_pose, H, W, f, ray0, n_rays, rays, stream);
    if (rc) return rc;
    // jitter is keyed by the GLOBAL pixel id, so the image does not depend on how it is sharded
    return nerf_amd_render_pixels_forward(rays, u, tbins, packed, precision, flags, seed, ray0, pixels,
                                          ws + align_up(n_rays * 24, 256), n_rays, N, stream);
}

int nerf_amd_sample_pdf(const float* ts, const float* w, const float* u, uint32_t flags, uint64_t seed,
                        int64_t ray_id0, float* ts_out, int64_t B, int Nc, int Nf, void* stream) {
    if (B < 0 || Nc <= 0 || Nf < 0) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (Nc < 3 || Nc > 256 || Nc + Nf > 512) return NERF_AMD_EUNSUP;
    if (!ts || !w || !ts_out) return NERF_AMD_EINVAL;
    if (flags & ~NERF_AMD_DEVICE_RNG) return NERF_AMD_EINVAL;
    if (!(flags & NERF_AMD_DEVICE_RNG) && !u && Nf > 0) return NERF_AMD_EINVAL;
    return nerf_amd_launch_sample_pdf(ts, w, u, ts_out, B, Nc, Nf, seed, ray_id0,
                                      (flags & NERF_AMD_DEVICE_RNG) ? 1 : 0, S(stream));
}

int64_t nerf_amd_render_hierarchical_workspace_bytes(int64_t n_rays, int Nc, int Nf) {
    if (n_rays < 0 || Nc <= 0 || Nf < 0) return NERF_AMD_EINVAL;
    // rays[n,6] + ts_c[n,Nc] + w_c[n,Nc] + ts_f[n,Nc+Nf]
    return align_up(n_rays * 24, 256) + 2 * align_up(n_rays * Nc * 4, 256) + align_up(n_rays * (int64_t)(Nc + Nf) * 4, 256);
}

int nerf_amd_render_hierarchical_forward(const float* h_pose, int H, int W, float f, int64_t ray0, int64_t n_rays,
                                         const float* u_c, const float* u_f, const float* tbins_c,
                                         void* packed_c, void* packed_f, int precision, uint32_t flags,
                                         uint64_t seed, float* pixels, void* workspace, int Nc, int Nf, void* stream) {
    if (n_rays < 0 || Nc <= 0 || Nf < 0 || bad_precision(precision)) return NERF_AMD_EINVAL;
    if (n_rays == 0) return 0;
    if (Nc < 3 || Nc > 256 || Nc + Nf > 512 || !fused_render(precision, Nc + Nf)) return NERF_AMD_EUNSUP;
    if (!workspace || !pixels || !packed_c || !packed_f || !tbins_c) return NERF_AMD_EINVAL;
    // explicit jitter for both passes, or the counter RNG with its seed in the argument: the seed-in-memory form would
    // need u_c to be that address, which this entry point does not offer (the coarse kernel would dereference it)
    if (flags & ~NERF_AMD_DEVICE_RNG) return NERF_AMD_EINVAL;
    if (!(flags & NERF_AMD_DEVICE_RNG) && (!u_c || (!u_f && Nf > 0))) return NERF_AMD_EINVAL;
    char* ws = reinterpret_cast<char*>(workspace);
    float* rays = reinterpret_cast<float*>(ws);
    ws += align_up(n_rays * 24, 256);
    float* ts_c = reinterpret_cast<float*>(ws);
    ws += align_up(n_rays * Nc * 4, 256);
    float* w_c = reinterpret_cast<float*>(ws);
    ws += align_up(n_rays * Nc * 4, 256);
    float* ts_f = reinterpret_cast<float*>(ws);
    int rc = nerf_amd_generate_rays(h_pose, H, W, f, ray0, n_rays, rays, stream);
    if (rc) return rc;
    // coarse pass: one fused launch that leaves only what the sampler needs (positions and weights)
    MlpArgs a{};
    a.rays = rays; a.u = u_c; a.tbins = tbins_c; a.packed = packed_c; a.ts_out = ts_c; a.w = w_c;
    a.P = n_rays * (int64_t)Nc; a.N = Nc; a.flags = flags; a.seed = seed; a.ray_id0 = ray0;
    rc = launch_mlp(a, 1, precision, S(stream));
    if (rc) return rc;
    rc = nerf_amd_launch_sample_pdf(ts_c, w_c, u_f, ts_f, n_rays, Nc, Nf, seed, ray0,
                                    (flags & NERF_AMD_DEVICE_RNG) ? 1 : 0, S(stream));
    if (rc) return rc;
    // fine pass on the merged, sorted positions -> clipped pixels
    return nerf_amd_render_pixels_forward(rays, ts_f, nullptr, packed_f, precision, NERF_AMD_TS_GIVEN, seed, ray0, pixels,
                                          nullptr, n_rays, Nc + Nf, stream);
}

int64_t nerf_amd_train_activation_bytes(int64_t P) {
    return P < 0 ? (int64_t)NERF_AMD_EINVAL : (int64_t)acts_total_bytes(P);
}

int nerf_amd_mlp_forward_train(const float* rays, const float* u, const float* tbins, void* packed,
                               uint32_t flags, uint64_t seed, int64_t ray_id0, float* raw, float* ts,
                               void* acts, int64_t B, int N, void* stream) {
    if (B < 0 || N <= 0) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (!rays || !packed || !raw || !acts) return NERF_AMD_EINVAL;
    if (bad_jitter(flags & ~NERF_AMD_STORE_E4M3, u, tbins)) return NERF_AMD_EINVAL;
    static_assert(NERF_AMD_STORE_E4M3 == NERF_FLAG_STORE_E4M3, "the flag travels to the kernel as it is");
    MlpArgs a{};
    a.rays = rays; a.u = u; a.tbins = tbins; a.packed = packed; a.raw = raw; a.ts_out = ts; a.acts = acts;
    a.P = B * (int64_t)N; a.N = N; a.flags = flags; a.seed = seed; a.ray_id0 = ray_id0;
    return nerf_amd_launch_mlp_bf16_16(&a, 1, S(stream));
}

int nerf_amd_mlp_forward_train_points(const float* pts, void* packed, float* out, void* acts, int64_t P,
                                      void* stream) {
    if (P < 0) return NERF_AMD_EINVAL;
    if (P == 0) return 0;
    if (!pts || !packed || !out || !acts) return NERF_AMD_EINVAL;
    MlpArgs a{};
    a.pts = pts; a.packed = packed; a.raw = out; a.acts = acts; a.P = P; a.N = 1;
    return nerf_amd_launch_mlp_bf16_16(&a, 1, S(stream));
}

int nerf_amd_encode_points_bf16(const float* pts, void* posx64, void* posd32, int64_t P, void* stream) {
    if (P < 0) return NERF_AMD_EINVAL;
    if (P == 0) return 0;
    if (!pts || !posx64 || !posd32) return NERF_AMD_EINVAL;
    MlpArgs a{};
    a.pts = pts; a.P = P; a.N = 1;
    return nerf_amd_launch_sample_encode_bf16(&a, posx64, posd32, S(stream));
}

int nerf_amd_mlp_backward(const float* d_raw, const void* bwd_image, const void* acts, void* dys, int64_t P,
                          void* stream) {
    if (P < 0) return NERF_AMD_EINVAL;
    if (P == 0) return 0;
    if (!d_raw || !bwd_image || !acts || !dys) return NERF_AMD_EINVAL;
    return nerf_amd_launch_mlp_backward(d_raw, bwd_image, acts, dys, P, 0, S(stream));
}

int64_t nerf_amd_train_activation_bytes_e4m3(int64_t P) {
    return P < 0 ? (int64_t)NERF_AMD_EINVAL : (int64_t)f8_acts_total_bytes(P);
}
int64_t nerf_amd_train_gradient_bytes_e4m3(int64_t P) {
    return P < 0 ? (int64_t)NERF_AMD_EINVAL : (int64_t)f8_data_bytes(P);
}
int64_t nerf_amd_param_gradients_scratch_e4m3_bytes(int64_t P) {
    return P < 0 ? (int64_t)NERF_AMD_EINVAL : (int64_t)nerf_amd_f8_scratch_bytes(P);
}
int nerf_amd_mlp_backward_e4m3(const float* d_raw, const void* bwd_image, const void* acts, void* dys, int64_t P,
                               void* stream) {
    if (P < 0) return NERF_AMD_EINVAL;
    if (P == 0) return 0;
    if (!d_raw || !bwd_image || !acts || !dys) return NERF_AMD_EINVAL;
    return nerf_amd_launch_mlp_backward(d_raw, bwd_image, acts, dys, P, 1, S(stream));
}
int nerf_amd_param_gradients_convert_e4m3(const void* posx64, const void* posd32, const void* scratch, void* scratch_e4m3,
                                          int64_t P, int which, void* stream) {
    if (P < 0 || which < 1 || which > 3) return NERF_AMD_EINVAL;
    if (P == 0) return 0;
    if (!scratch_e4m3 || ((which & 1) && (!posx64 || !posd32)) || ((which & 2) && !scratch)) return NERF_AMD_EINVAL;
    return nerf_amd_launch_param_gradients_convert_e4m3(posx64, posd32, scratch, scratch_e4m3, P, which, S(stream));
}
int nerf_amd_param_gradients_finish_e4m3(const void* acts, const void* dys, const void* scratch_e4m3, float* grads, int64_t P,
                                         int bucket, void* stream) {
    if (P < 0 || !grads || bucket < 0 || bucket > 2) return NERF_AMD_EINVAL;
    if (P > 0 && (!acts || !dys || !scratch_e4m3)) return NERF_AMD_EINVAL;
    return nerf_amd_launch_param_gradients_finish_e4m3(acts, dys, scratch_e4m3, grads, P, bucket, S(stream));
}

int nerf_amd_sample_encode_bf16(const float* rays, const float* u, const float* tbins, uint32_t flags,
                                uint64_t seed, int64_t ray_id0, void* posx64, void* posd32, float* ts, int64_t B,
                                int N, void* stream) {
    if (B < 0 || N <= 0) return NERF_AMD_EINVAL;
    if (B == 0) return 0;
    if (!rays || !posx64 || !posd32) return NERF_AMD_EINVAL;
    if (bad_jitter(flags, u, tbins)) return NERF_AMD_EINVAL;
    MlpArgs a{};
    a.rays = rays; a.u = u; a.tbins = tbins; a.ts_out = ts;
    a.P = B * (int64_t)N; a.N = N; a.flags = flags; a.seed = seed; a.ray_id0 = ray_id0;
    return nerf_amd_launch_sample_encode_bf16(&a, posx64, posd32, S(stream));
}

int64_t nerf_amd_param_gradients_scratch_bytes(int64_t P) { return P < 0 ? (int64_t)NERF_AMD_EINVAL : P * 64; }

int nerf_amd_param_gradients(const float* d_raw, const void* acts, const void* dys, const void* posx64,
                             const void* posd32, void* scratch, float* grads, int64_t P, void* stream) {
    if (P < 0 || !grads) return NERF_AMD_EINVAL;
    if (P > 0 && (!d_raw || !acts || !dys || !posx64 || !posd32 || !scratch)) return NERF_AMD_EINVAL;
    return nerf_amd_launch_param_gradients(d_raw, acts, dys, posx64, posd32, scratch, grads, P, S(stream));
}

int nerf_amd_param_gradients_begin(const float* d_raw, void* scratch, float* grads, int64_t P, void* stream) {
    if (P < 0 || !grads) return NERF_AMD_EINVAL;
    if (P > 0 && (!d_raw || !scratch)) return NERF_AMD_EINVAL;
    return nerf_amd_launch_param_gradients_begin(d_raw, scratch, grads, P, S(stream));
}

int nerf_amd_param_gradients_finish(const void* acts, const void* dys, const void* posx64, const void* posd32,
                                    const void* scratch, float* grads, int64_t P, void* stream) {
    return nerf_amd_param_gradients_finish_bucket(acts, dys, posx64, posd32, scratch, grads, P, 0, stream);
}

int nerf_amd_param_gradients_finish_bucket(const void* acts, const void* dys, const void* posx64, const void* posd32,
                                           const void* scratch, float* grads, int64_t P, int bucket, void* stream) {
    if (P < 0 || !grads || bucket < 0 || bucket > 2) return NERF_AMD_EINVAL;
    if (P > 0 && (!acts || !dys || !posx64 || !posd32 || !scratch)) return NERF_AMD_EINVAL;
    return nerf_amd_launch_param_gradients_finish(acts, dys, posx64, posd32, scratch, grads, P, bucket, S(stream));
}

int nerf_amd_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, int64_t step, void* stream) {
    if (n < 0 || step < 1) return NERF_AMD_EINVAL;
    if (n == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return NERF_AMD_EINVAL;
    // bias corrections in double on the host, as torch does for a python-number step
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    return nerf_amd_launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1,
                                (float)sqrt(bc2), S(stream));
}

int nerf_amd_mt19937_uniform(const uint32_t* state624, int next, float* out, int64_t n, uint32_t* state_out624,
                             void* stream) {
    if (n < 0 || next < 0 || next > 624) return NERF_AMD_EINVAL;
    if (!state624 || !state_out624 || (n > 0 && !out)) return NERF_AMD_EINVAL;
    return nerf_amd_launch_mt19937_uniform(state624, next, out, n, state_out624, S(stream));
}

int64_t nerf_amd_mt19937_segments(int next, int64_t n, int64_t seg_words) {
    if (n < 0 || next < 0 || next > 624 || seg_words <= 0 || seg_words % 624) return NERF_AMD_EINVAL;
    const int64_t avail = 624 - next;
    return n > avail + seg_words ? 1 + (n - avail - seg_words + seg_words - 1) / seg_words : 1;
}

int nerf_amd_mt19937_uniform_par(const uint32_t* state624, int next, float* out, int64_t n, uint32_t* state_out624,
                                 const uint32_t* polys, int levels, int64_t seg_words, uint32_t* seg_states,
                                 void* stream) {
    if (n < 0 || next < 0 || next > 624 || levels < -4096 || levels > 30) return NERF_AMD_EINVAL;
    if (seg_words <= 0 || seg_words % 624) return NERF_AMD_EINVAL;
    if (!state624 || !state_out624 || (n > 0 && !out)) return NERF_AMD_EINVAL;
    const int64_t nseg = nerf_amd_mt19937_segments(next, n, seg_words);
    if (nseg > 1 && (!polys || !seg_states)) return NERF_AMD_EINVAL;
    if (levels >= 0 ? nseg > ((int64_t)1 << levels) : nseg > 1 - (int64_t)levels) return NERF_AMD_EUNSUP;
    return nerf_amd_launch_mt19937_uniform_par(state624, next, out, n, state_out624, polys, levels, seg_words, seg_states,
                                               S(stream));
}

int nerf_amd_mt19937_raw(const uint32_t* state624, int next, uint32_t* out, int64_t n, uint32_t* state_out624, void* stream) {
    if (n < 0 || next < 0 || next > 624) return NERF_AMD_EINVAL;
    if (!state624 || (n > 0 && !out)) return NERF_AMD_EINVAL;
    return nerf_amd_launch_mt19937_raw(state624, next, out, n, state_out624, S(stream));
}

int nerf_amd_mt19937_jump_poly(int64_t blocks, const uint32_t* h_phi624, uint32_t* h_poly624) {
    if (blocks < 0 || blocks > ((int64_t)1 << 52) || !h_phi624 || !h_poly624) return NERF_AMD_EINVAL;
    return nerf_amd_host_mt19937_jump_poly(blocks, h_phi624, h_poly624) == 0 ? 0 : NERF_AMD_EINVAL;
}

int nerf_amd_mt19937_advance(const uint32_t* state624, const uint32_t* poly624, uint32_t* state_out624, void* stream) {
    if (!state624 || !poly624 || !state_out624 || state624 == state_out624) return NERF_AMD_EINVAL;
    return nerf_amd_launch_mt19937_advance(state624, poly624, state_out624, S(stream));
}

int nerf_amd_mt19937_uniform_after(const uint32_t* state624, const uint32_t* polys, int segments, int next_after, float* out,
                                   int64_t n, uint32_t* state_out624, int64_t seg_words, uint32_t* seg_states, void* stream) {
    if (n <= 0 || segments < 1 || segments > 4096 || next_after < 0 || next_after > 624) return NERF_AMD_EINVAL;
    if (seg_words <= 0 || seg_words % 624) return NERF_AMD_EINVAL;
    if (!state624 || !polys || !out || !state_out624 || !seg_states) return NERF_AMD_EINVAL;
    if (nerf_amd_mt19937_segments(next_after, n, seg_words) != segments) return NERF_AMD_EINVAL;
    return nerf_amd_launch_mt19937_uniform_after(state624, polys, segments, next_after, out, n, state_out624, seg_words, seg_states,
                                                 S(stream));
}

int64_t nerf_amd_select_workspace_bytes(int64_t B) { return B < 0 ? NERF_AMD_EINVAL : align_up(B * 12 + 16, 256); }

int nerf_amd_select_rays(const uint32_t* draws, uint64_t seed, const uint64_t* seed_mem, int64_t n, int64_t B,
                         const float* table, const float* colours, float* rays_out, float* gt_out, int64_t* ids_out,
                         void* workspace, void* stream) {
    if (n < 0 || B < 0 || B > n) return NERF_AMD_EINVAL;
    if (n >= (int64_t)(0xffffffffu / 20u)) return NERF_AMD_EUNSUP;      // torch.randperm switches algorithm there
    if (B == 0) return 0;
    if (!workspace || (rays_out && !table) || (gt_out && !colours)) return NERF_AMD_EINVAL;
    if (((uintptr_t)table & 7) || ((uintptr_t)rays_out & 7) || ((uintptr_t)seed_mem & 7) || ((uintptr_t)workspace & 15)) return NERF_AMD_EINVAL;
    return nerf_amd_launch_select_rays(draws, seed, reinterpret_cast<const unsigned long long*>(seed_mem), n, B, table, colours,
                                       rays_out, gt_out, reinterpret_cast<long long*>(ids_out), workspace, S(stream));
}

int nerf_amd_linear_f32(const float* A, int64_t sa_i, int64_t sa_k, const float* A_mask, const float* B, int64_t sb_k,
                        int64_t sb_j, const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, uint32_t flags,
                        void* stream) {
    if (M < 0 || N < 0 || K < 0 || (flags & ~3u)) return NERF_AMD_EINVAL;
    if (M == 0 || N == 0) return 0;
    if (!C || ldc < N || (K > 0 && (!A || !B))) return NERF_AMD_EINVAL;
    return nerf_amd_launch_linear_f32(A, sa_i, sa_k, A_mask, B, sb_k, sb_j, bias, C, ldc, M, N, K, (int)flags, S(stream));
}

int64_t nerf_amd_pinned_device_address(const void* host) {
    if (!host) return NERF_AMD_EINVAL;
    (void)hipGetLastError();
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void*>(host), 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        return NERF_AMD_EINVAL;                    // not pinned (or not mapped into the device's address space)
    }
    return (int64_t)reinterpret_cast<uintptr_t>(d);
}

int nerf_amd_hyper_fetch(const float* ring_dev, int slots, float* hyper, uint32_t* counter, void* stream) {
    if (!ring_dev || !hyper || !counter || slots <= 0) return NERF_AMD_EINVAL;
    return nerf_amd_launch_hyper_fetch(ring_dev, slots, hyper, counter, S(stream));
}

int nerf_amd_adam_step_hyper(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                             const float* hyper, void* stream) {
    if (n < 0) return NERF_AMD_EINVAL;
    if (n == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !hyper) return NERF_AMD_EINVAL;
    return nerf_amd_launch_adam_hyper(params, grads, exp_avg, exp_avg_sq, n, hyper, S(stream));
}

}  // extern "C"
