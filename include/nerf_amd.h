/* nerf_amd.h -- C ABI of libnerf_amd.so: the MI355X (gfx950) NeRF render hot path.
 *
 * Drop-in boundary for the hot path of UCSD-Comp-Imaging/Nerf-Simple.  The
 * reference has no FFI; its boundary is plain Python call signatures in
 * package `utils`.  Each entry point below names the
 * reference function (file:line, relative to the reference repo) whose body it
 * replaces; the Python host side (nerf-simple_amd/utils/) keeps the reference's
 * names, argument order, defaults and return order and calls these through
 * ctypes (binding shown in INTEGRATION.md).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless its name starts with `h_`;
 *     fp32, row-major, contiguous; never written unless documented as output
 *     (one exception, which is why `packed` is not const where a 16-bit MLP kernel may run: the
 *     sticky status word behind a packed 16-bit image, see nerf_amd_packed_status_offset);
 *   - the library allocates nothing, frees nothing and keeps no pointer after
 *     return; workspaces are caller-provided;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it,
 *     do no host synchronisation and are safe to capture into a hipGraph;
 *   - return value: 0 = launched, otherwise a negative NERF_AMD_E* code or a
 *     positive hipError_t; nothing throws or exits across the ABI.
 */
#ifndef NERF_AMD_H
#define NERF_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_AMD_ABI_VERSION 5

/* error codes */
#define NERF_AMD_EINVAL   (-1)   /* bad argument (null pointer, negative size, ...) */
#define NERF_AMD_EUNSUP   (-2)   /* unsupported configuration */

/* precision of the fused MLP */
#define NERF_AMD_F32   0   /* exact-f32 MFMA (v_mfma_f32_16x16x4_f32), fp32 end to end */
#define NERF_AMD_BF16  1   /* bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulate (the flagship) */
#define NERF_AMD_BF16_BWD 3 /* nerf_amd_pack_weights / nerf_amd_packed_bytes only: the transposed image of nerf_amd_mlp_backward */
#define NERF_AMD_FP16  2   /* fp16 operands on v_mfma_f32_16x16x32_f16: same rate, 11-bit mantissa; range 65504 */

/* flags of nerf_amd_render_forward / nerf_amd_mlp_forward_rays */
#define NERF_AMD_TS_GIVEN   1u  /* `u` holds sample positions ts[B,N], not jitter */
#define NERF_AMD_DEVICE_RNG 2u  /* `u` ignored (may be NULL): jitter from the counter RNG */
#define NERF_AMD_SEED_IN_MEMORY 4u /* with NERF_AMD_DEVICE_RNG: `u` is the DEVICE ADDRESS of a uint64 that is added to
                                    * `seed` when the kernel runs -- a launch captured into a hipGraph is replayed with
                                    * frozen arguments, and this is how every replay of a training step draws fresh
                                    * jitter (the reference draws torch.rand(B,N) anew per call, utils/rendering.py:28) */

/* ---- introspection (host only, no GPU needed) ------------------------------ */
int      nerf_amd_abi_version(void);
/* 595844 = parameters of Nerf(Lp=10, Ld=4, H=256), reference utils/nets.py:9-32 */
int64_t  nerf_amd_param_count(void);
/* bytes of the packed weight image for a precision (the caller allocates it) */
int64_t  nerf_amd_packed_bytes(int precision);
/* Range guard of the 16-bit images.  The reference is fp32 with no range limit (utils/nets.py:16-32); fp16
 * operands overflow beyond 65504.  A packed NERF_AMD_FP16 / NERF_AMD_BF16 image ends with a 256-byte status
 * block of sticky uint32 flags (0 / 1), zeroed by nerf_amd_pack_weights:
 *   word NERF_AMD_STATUS_WORD_NONFINITE     set by every 16-bit MLP forward kernel when a point's (rgb, sigma) output
 *                                           is inf or NaN (where an overflowed hidden activation ends up);
 *   word NERF_AMD_STATUS_WORD_WEIGHT_RANGE  set by the packer if a weight is not finite in the operand type (beyond its
 *                                           range, or NaN / inf to begin with).
 * Returns the byte offset of word 0 inside the image, -1 for images without a status block (NERF_AMD_F32,
 * NERF_AMD_BF16_BWD).  The host wrapper (utils/nets.py) reads it after the first fp16 render of a weight set and
 * falls back to bf16 operands with a warning instead of returning NaN pixels. */
#define NERF_AMD_STATUS_WORD_NONFINITE     0
#define NERF_AMD_STATUS_WORD_WEIGHT_RANGE  1
int64_t  nerf_amd_packed_status_offset(int precision);
/* bytes of workspace nerf_amd_render_forward / _pixels_forward need for B rays x N samples.
 * 0 up to 768 samples per ray: those renders run as ONE launch (sampling + encoding + MLP +
 * compositing, samples composited out of LDS) and `workspace` may be NULL; longer rays take the
 * two-launch path through raw[B,N,4] + ts[B,N]. */
int64_t  nerf_amd_render_workspace_bytes(int precision, int64_t B, int N);
/* host-side self-check of the packed-weight index math (bijectivity of the
 * k-permutations, offsets, sizes); 0 = consistent.  Used by the CPU tests. */
int      nerf_amd_layout_selfcheck(void);
/* host-side: source column (or -1 = padding) that the packed image holds at
 * (layer, k-step, lane group 0..3, element 0..7 [16-bit images] / 0 [f32]); -2 = out of range.
 * Lets tests audit the packing. */
int      nerf_amd_layout_src_col(int precision, int layer, int kstep, int group, int elem);

/* ---- weights ---------------------------------------------------------------- */
/* Pack the 24 state-dict tensors of the reference Nerf (utils/nets.py:16-32),
 * given flattened in state_dict order as one fp32 vector `params`[595844], into
 * the MFMA-fragment-ordered image the fused kernels stream.  Derived cache:
 * re-run after any parameter update.  */
int nerf_amd_pack_weights(const float* params, void* packed, int precision, void* stream);
/* The two images a training step needs (NERF_AMD_BF16 and NERF_AMD_BF16_BWD) in one launch.  Clears the status block of
 * the forward image except NERF_AMD_STATUS_WORD_WEIGHT_RANGE, which it only ever sets (a non-finite weight stays
 * non-finite under Adam; nerf_amd_pack_weights clears the word): `packed_bf16` is an image that nerf_amd_pack_weights
 * has filled before (any weights), re-packed in place from then on. */
int nerf_amd_pack_weights_train(const float* params, void* packed_bf16, void* packed_bwd, void* stream);

/* ---- positional encoding: utils/xyz.py:6-36 --------------------------------- */
/* gamma(x, L): x[n] -> out[n, 2L] = [sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]
 * (utils/xyz.py:6-14).  `x_stride` in floats lets x be a column of a wider table. */
int nerf_amd_gamma(const float* x, int64_t x_stride, float* out, int64_t n, int L, void* stream);
/* positional_encoder(vec, Lp, Ld): vec[P,6] -> posx[P,3+6Lp], posd[P,3+6Ld]
 * (utils/xyz.py:16-36), columns grouped per coordinate. */
int nerf_amd_positional_encoder(const float* vec, float* posx, float* posd,
                                int64_t P, int Lp, int Ld, void* stream);

/* ---- the MLP: Nerf.forward, utils/nets.py:34-43 ------------------------------ */
/* pts[P,6] = [x,y,z,d1,d2,d3] -> out[P,4] = [r,g,b,sigma] (raw: no sigmoid, no
 * softplus).  Encoding + 12 dense layers fused in one kernel; activations never
 * leave the CU.  `packed` from nerf_amd_pack_weights with the same precision. */
int nerf_amd_mlp_forward(const float* pts, void* packed, float* out,
                         int64_t P, int precision, void* stream);

/* ---- sampling + query points on their own: utils/rendering.py:24-40 ------------ */
/* rays[B,6] (+ u / ts / device RNG and tbins as in nerf_amd_render_forward) -> query_pts[B*N,6] =
 * [origin + direction * t, direction / ||direction||] ray-major / sample-minor, and ts[B,N] (may be NULL).
 * For a caller whose network is not the fused one (render_nerf with a foreign `net`): that net's own forward
 * runs on query_pts, nerf_amd_volume_render_rays composites. */
int nerf_amd_query_points(const float* rays, const float* u, const float* tbins,
                          uint32_t flags, uint64_t seed, int64_t ray_id0,
                          float* query_pts, float* ts, int64_t B, int N, void* stream);

/* ---- compositing: volume_render, utils/rendering.py:47-85 -------------------- */
/* raw[B,N,4], ts[B,N], dirs[B] (3 floats at stride `dirs_stride` floats) ->
 * rgb[B,3], disp[B], alpha[B,N], acc[B], w[B,N].  alpha and w may be NULL.
 * One wavefront per ray; transmittance by a wave-level product scan.
 * N == 1 reproduces the reference's degenerate result (every compositing entry point and the fused renders):
 * its delta construction (utils/rendering.py:60-61) leaves the sample axis EMPTY there, so rgb = acc = 0,
 * disparity = NaN, alpha / w (shape [B,0] in the reference) are not written and d_raw is zero. */
int nerf_amd_volume_render(const float* raw, const float* ts,
                           const float* dirs, int64_t dirs_stride,
                           float* rgb, float* disp, float* alpha, float* acc, float* w,
                           int64_t B, int N, void* stream);

/* Image-driver form of the compositor (stage 2 of nerf_amd_render_image_forward
 * on its own): raw[B,N,4], ts[B,N] from nerf_amd_mlp_forward_rays and the same
 * rays[B,6] -> pixels[B,4] = [clip(rgb,0,1), disparity]  (utils/rendering.py:102-105). */
int nerf_amd_volume_render_pixels(const float* raw, const float* ts, const float* rays,
                                  float* pixels, int64_t B, int N, void* stream);

/* Backward of the above: d loss / d raw [B,N,4] from the upstream gradients of
 * the five outputs (any of g_* may be NULL = zero).  Autograd through
 * volume_render in the training step, reference train.py:51-54.  ts and dirs
 * get no gradient (they carry none in the reference either).  N <= 512. */
int nerf_amd_volume_render_backward(const float* raw, const float* ts,
                                    const float* dirs, int64_t dirs_stride,
                                    const float* g_rgb, const float* g_disp, const float* g_alpha,
                                    const float* g_acc, const float* g_w,
                                    float* d_raw, int64_t B, int N, void* stream);

/* Both with the directions taken from rays[B,6] and normalised inside the kernel,
 * dirs = rays[:,3:] / ||rays[:,3:]|| as render_nerf does (utils/rendering.py:37,43):
 * no [B,3] temporary, no extra launch (the training step uses these). */
int nerf_amd_volume_render_rays(const float* raw, const float* ts, const float* rays,
                                float* rgb, float* disp, float* alpha, float* acc, float* w,
                                int64_t B, int N, void* stream);
int nerf_amd_volume_render_rays_backward(const float* raw, const float* ts, const float* rays,
                                         const float* g_rgb, const float* g_disp, const float* g_alpha,
                                         const float* g_acc, const float* g_w,
                                         float* d_raw, int64_t B, int N, void* stream);

/* Training form (reference train.py:51-54 between the MLP forward and its backward): compositing
 * forward, loss = MSELoss(rgb, target) (mean over 3B elements), and the backward of both in ONE launch:
 * raw, ts, rays, target[B,3] -> rgb[B,3] (may be NULL; feed it to nerf_amd_mse_loss for the loss value)
 * and d_raw[B,N,4] = d loss / d raw.  N <= 512. */
int nerf_amd_volume_render_mse_backward(const float* raw, const float* ts, const float* rays,
                                        const float* target, float* rgb, float* d_raw,
                                        int64_t B, int N, void* stream);

/* ---- the whole path: render_nerf, utils/rendering.py:13-45 ------------------- */
/* rays[B,6] = [origin, direction] -> (rgb[B,3], disp[B], alpha[B,N], acc[B], w[B,N]).
 *   u        jitter in [0,1) [B,N] exactly as the reference draws it with
 *            torch.rand(B,N) (utils/rendering.py:28); or ts[B,N] with
 *            NERF_AMD_TS_GIVEN; or unused with NERF_AMD_DEVICE_RNG (counter RNG
 *            keyed by (seed, ray_id0 + ray, sample) so results do not depend on
 *            batching or sharding).
 *   tbins    linspace(tn, tf, N+1) computed by the caller [N+1] (device), so
 *            bin edges are bit-identical to torch.linspace (utils/rendering.py:25)
 *   alpha,w  optional (NULL to skip the 8 B/sample of output traffic)
 *   workspace  nerf_amd_render_workspace_bytes(precision,B,N) bytes, 256-B aligned (NULL if 0) */
int nerf_amd_render_forward(const float* rays, const float* u, const float* tbins,
                            void* packed, int precision, uint32_t flags,
                            uint64_t seed, int64_t ray_id0,
                            float* rgb, float* disp, float* alpha, float* acc, float* w,
                            void* workspace, int64_t B, int N, void* stream);

/* Image-driver form (the body of utils/rendering.py:102-105 for one batch): the same render, output
 * pixels[B,4] = [clip(rgb,0,1), disparity]; rgb is clipped AFTER compositing, disparity is not. */
int nerf_amd_render_pixels_forward(const float* rays, const float* u, const float* tbins,
                                   void* packed, int precision, uint32_t flags,
                                   uint64_t seed, int64_t ray_id0,
                                   float* pixels, void* workspace, int64_t B, int N, void* stream);

/* Stage 1 of the above on its own (sampling + encoding + MLP): writes
 * raw[B,N,4] and ts[B,N].  Exposed for the importance-sampling caller, which
 * needs explicit ts. */
int nerf_amd_mlp_forward_rays(const float* rays, const float* u, const float* tbins,
                              void* packed, int precision, uint32_t flags,
                              uint64_t seed, int64_t ray_id0,
                              float* raw, float* ts, int64_t B, int N, void* stream);

/* ---- image drivers: render_poses / render_image, utils/rendering.py:88-153 ---- */
/* Pinhole rays on the device (utils/xyz.py:38-52 + utils/rendering.py:129-134):
 * rays[i] = [pose[:3,3], pose[:3,:3] @ ((w-W//2)/f, -(h-H//2)/f, -1)] for pixel
 * p = ray0 + i = h*W + w.  h_pose: HOST pointer to a row-major 3x4 / 4x4 pose. */
int nerf_amd_generate_rays(const float* h_pose, int H, int W, float f,
                           int64_t ray0, int64_t n_rays, float* rays, void* stream);
int64_t nerf_amd_render_image_workspace_bytes(int precision, int64_t n_rays, int N);
/* One call = the body of the reference's per-image loop (utils/rendering.py:139-151)
 * for pixels [ray0, ray0+n_rays) of an HxW view: ray generation, render_nerf,
 * clip(rgb,0,1) after compositing, disparity un-clipped ->
 * pixels[n_rays,4] = [r,g,b,disparity].  Two launches (three on the two-launch path), no host sync; the
 * multi-GPU driver calls it per rank and all-gathers `pixels`.  u / tbins /
 * flags / seed as in nerf_amd_render_forward (u indexed from ray0). */
int nerf_amd_render_image_forward(const float* h_pose, int H, int W, float f,
                                  int64_t ray0, int64_t n_rays,
                                  const float* u, const float* tbins,
                                  void* packed, int precision, uint32_t flags, uint64_t seed,
                                  float* pixels, void* workspace, int N, void* stream);

/* ---- hierarchical sampling (BASELINE config 4) -------------------------------- */
/* ABSENT from the reference (README.md:3, configs/lego.yaml:7): parity unpinned.
 * Inverse-CDF placement of Nf new samples from the coarse pass's weights (the
 * NeRF paper's sample_pdf over interior bins), merged and sorted with the Nc
 * coarse positions: ts[B,Nc], w[B,Nc], u[B,Nf] in [0,1) (or NERF_AMD_DEVICE_RNG)
 * -> ts_out[B,Nc+Nf] ascending.  Feed ts_out to nerf_amd_render_forward with
 * NERF_AMD_TS_GIVEN for the fine pass.  3 <= Nc <= 256, Nc+Nf <= 512. */
int nerf_amd_sample_pdf(const float* ts, const float* w, const float* u,
                        uint32_t flags, uint64_t seed, int64_t ray_id0,
                        float* ts_out, int64_t B, int Nc, int Nf, void* stream);

/* BASELINE config 4 as ONE call for pixels [ray0, ray0+n_rays) of an HxW view: device ray generation ->
 * coarse render (Nc stratified samples, network `packed_c`; only its positions and weights are kept)
 * -> nerf_amd_sample_pdf -> fine render of `packed_f` on the Nc+Nf merged positions ->
 * pixels[n_rays,4] = [clip(rgb,0,1), disparity].  Four launches, no host sync, nothing per-sample but
 * ts / w of the coarse pass and ts of the fine pass touches HBM.  u_c[n,Nc] / u_f[n,Nf] explicit
 * uniforms, or NERF_AMD_DEVICE_RNG (both keyed by the global pixel id: sharding-invariant).
 * 3 <= Nc <= 256, Nc+Nf <= 512; otherwise NERF_AMD_EUNSUP (compose the three stages instead).
 * Parity unpinned like nerf_amd_sample_pdf (the reference has no hierarchical sampling). */
int64_t nerf_amd_render_hierarchical_workspace_bytes(int64_t n_rays, int Nc, int Nf);
int nerf_amd_render_hierarchical_forward(const float* h_pose, int H, int W, float f,
                                         int64_t ray0, int64_t n_rays,
                                         const float* u_c, const float* u_f, const float* tbins_c,
                                         void* packed_c, void* packed_f, int precision,
                                         uint32_t flags, uint64_t seed,
                                         float* pixels, void* workspace, int Nc, int Nf, void* stream);

/* Training-side front end: sampling + point assembly + encoding in one launch
 * (utils/rendering.py:24-40 + utils/xyz.py:16-36): rays[B,6] (+ u / ts / device
 * RNG as in nerf_amd_render_forward) -> posx[B*N,63], posd[B*N,27], ts[B,N]
 * (Lp = 10, Ld = 4), fp32, reference column order. */
int nerf_amd_sample_encode(const float* rays, const float* u, const float* tbins,
                           uint32_t flags, uint64_t seed, int64_t ray_id0,
                           float* posx, float* posd, float* ts, int64_t B, int N, void* stream);

/* ---- fused training path of the dense layers (reference train.py:51-54) --------
 * bf16 ONLY: the fused training kernels exist in bf16; a caller asking them for another precision gets
 * NERF_AMD_EUNSUP from its host wrapper.  (Exact fp32 training is the layer-by-layer composition of
 * nerf_amd_positional_encoder, nerf_amd_linear_f32 and the compositor's backward: at the end of this header.)
 * Forward as nerf_amd_mlp_forward_rays (bf16) that ALSO saves every layer's output for the
 * weight gradients (bf16; L0..L7 post-ReLU 256 features, L8 = the linear 256->256, L9 = colour
 * hidden 128; P = B*N points) in the point-blocked layout the kernels write and read with
 * contiguous 256-byte runs: layer L at L * ceil(P/256) * 128 KiB, inside it tile t (256 points)
 * at t * 128 KiB as [feature/8 (32)][point in tile (256)][8 bf16].  Then, at
 * 10 * ceil(P/256) * 128 KiB, the ReLU masks for the dX chain: one bit per feature,
 * 10 * ceil(P/256) * 8 KiB, in the kernels' register order.  (Both layouts: csrc/nerf_layout.h;
 * decoded in tests/test_gpu_training.py.) */
int64_t nerf_amd_train_activation_bytes(int64_t P);
int nerf_amd_mlp_forward_train(const float* rays, const float* u, const float* tbins,
                               void* packed_bf16, uint32_t flags, uint64_t seed, int64_t ray_id0,
                               float* raw, float* ts, void* acts, int64_t B, int N, void* stream);
/* The same for explicit points, i.e. Nerf.forward(v) under autograd (utils/nets.py:34-43):
 * pts[P,6] -> out[P,4], activations saved as above; and the matching bf16 encoder rows. */
int nerf_amd_mlp_forward_train_points(const float* pts, void* packed_bf16, float* out,
                                      void* acts, int64_t P, void* stream);
int nerf_amd_encode_points_bf16(const float* pts, void* posx64, void* posd32, int64_t P, void* stream);
/* Backward dX chain: d_raw[P,4] (from nerf_amd_volume_render_backward) + the ReLU
 * mask bits inside `acts` (the bf16 activations themselves are not read here) ->
 * dys: every layer's pre-activation gradient, bf16, point-blocked like the bf16 part
 * of `acts`.  The gradient w.r.t. activations stays on-chip between
 * layers.  `bwd_image` from nerf_amd_pack_weights(..., NERF_AMD_BF16_BWD).  Weight
 * gradients are then dW_L = dys[L]^T @ input_L: nerf_amd_param_gradients. */
int nerf_amd_mlp_backward(const float* d_raw, const void* bwd_image, const void* acts,
                          void* dys, int64_t P, void* stream);

/* Encoder outputs as the dW GEMM wants them: bf16, posx64[P,64] (col 63 zero),
 * posd32[P,32] (cols 27..31 zero); otherwise as nerf_amd_sample_encode. */
int nerf_amd_sample_encode_bf16(const float* rays, const float* u, const float* tbins,
                                uint32_t flags, uint64_t seed, int64_t ray_id0,
                                void* posx64, void* posd32, float* ts, int64_t B, int N, void* stream);
/* All 24 parameter gradients of the step in ONE flat fp32 vector `grads`[595844]
 * (state_dict order; zeroed by the call): dW_L = dys[L]^T @ input_L as split-K
 * GEMMs over the points, db_L = column sums.  The vector is also the bucket of the
 * data-parallel all-reduce.  scratch: nerf_amd_param_gradients_scratch_bytes(P). */
int64_t nerf_amd_param_gradients_scratch_bytes(int64_t P);
int nerf_amd_param_gradients(const float* d_raw, const void* acts, const void* dys,
                             const void* posx64, const void* posd32, void* scratch,
                             float* grads, int64_t P, void* stream);

/* The same in two parts, so a captured step can overlap the first with the dX chain (both need only
 * d_raw): _begin zeroes `grads`, packs d_raw into `scratch` and adds the two head bias gradients;
 * _finish runs the 14 products and the other bias sums. */
int nerf_amd_param_gradients_begin(const float* d_raw, void* scratch, float* grads, int64_t P, void* stream);
int nerf_amd_param_gradients_finish(const void* acts, const void* dys, const void* posx64, const void* posd32,
                                    const void* scratch, float* grads, int64_t P, void* stream);
/* Data-parallel training reduces `grads` in two buckets so the exchange overlaps the arithmetic: bucket 1 =
 * skip_conn_layer ... color_fc.2 (the tail of the vector, whose products run first), bucket 2 = layers_0.* (its
 * head); bucket 0 = everything (= nerf_amd_param_gradients_finish).  nerf_amd_grad_bucket_range gives a bucket's
 * [first, first + count) inside the flat vector.  The caller all-reduces bucket 1 while bucket 2 is computed. */
int nerf_amd_grad_bucket_range(int bucket, int64_t* h_first, int64_t* h_count);
int nerf_amd_param_gradients_finish_bucket(const void* acts, const void* dys, const void* posx64, const void* posd32,
                                           const void* scratch, float* grads, int64_t P, int bucket, void* stream);

/* ---- the same three steps with the saved tensors in the 8-bit storage form ---------------------------------------
 * What the forward saves and the dX chain writes is read by the weight gradients only: sums over all P points
 * (reference train.py:54, loss.backward()).  With NERF_AMD_STORE_E4M3 both buffers hold OCP e4m3 bytes plus one
 * power-of-two exponent per (32 features x 32 points) instead of bf16 -- half the bytes of the step's three
 * HBM-bound kernels; the chain itself still runs on bf16 values in registers, so raw / d_raw / the masks are bit for
 * bit those of the bf16 form, and the gradients differ by the operand rounding of the products only (bounded in
 * tests/test_gpu_storage.py against the reference's minibatch deviation).
 * Layout (csrc/nerf_layout.h): layer L at L * ceil(P/256) * 64 KiB, tile t at t * 64 KiB as
 * [feature/16 (16)][point in tile (256)][16 x e4m3]; behind the 10 layers, per layer and 32-point block 8 exponent
 * bytes (byte Q: features 32Q .. 32Q+31; value = e4m3 * 2^(byte - 127); the producers choose one exponent per 128
 * features, so bytes 4k .. 4k+3 are equal -- a consumer need not rely on it); `acts` then carries the ReLU masks.
 *   nerf_amd_mlp_forward_train(..., flags | NERF_AMD_STORE_E4M3, ...)  with acts of nerf_amd_train_activation_bytes_e4m3(P)
 *   nerf_amd_mlp_backward_e4m3: dys of nerf_amd_train_gradient_bytes_e4m3(P)
 *   nerf_amd_param_gradients_begin (unchanged: packs d_raw into `scratch`);
 *   nerf_amd_param_gradients_convert_e4m3: the narrow operands of the products into `scratch_e4m3`
 *   (nerf_amd_param_gradients_scratch_e4m3_bytes(P)) -- which & 1: the bf16 encoder rows posx64 / posd32 (needs only
 *   nerf_amd_sample_encode_bf16: a captured step runs it beside the forward), which & 2: the packed d_raw of `scratch`
 *   (needs ..._begin: beside the dX chain);
 *   nerf_amd_param_gradients_finish_e4m3: the 14 products on the block-scaled 8-bit MFMA and the bias sums (buckets as in
 *   nerf_amd_param_gradients_finish_bucket). */
#define NERF_AMD_STORE_E4M3 8u /* nerf_amd_mlp_forward_train only */
int64_t nerf_amd_train_activation_bytes_e4m3(int64_t P);
int64_t nerf_amd_train_gradient_bytes_e4m3(int64_t P);
int64_t nerf_amd_param_gradients_scratch_e4m3_bytes(int64_t P);
int nerf_amd_mlp_backward_e4m3(const float* d_raw, const void* bwd_image, const void* acts_e4m3,
                               void* dys_e4m3, int64_t P, void* stream);
int nerf_amd_param_gradients_convert_e4m3(const void* posx64, const void* posd32, const void* scratch,
                                          void* scratch_e4m3, int64_t P, int which, void* stream);
int nerf_amd_param_gradients_finish_e4m3(const void* acts_e4m3, const void* dys_e4m3, const void* scratch_e4m3,
                                         float* grads, int64_t P, int bucket, void* stream);

/* ---- loss: nn.MSELoss(), reference train.py:42,52 -------------------------------- */
/* loss[0] = mean((pred - target)^2) over n elements; g_pred[n] (may be NULL) =
 * d loss / d pred = 2 (pred - target) / n.  One workgroup, fixed summation order. */
int nerf_amd_mse_loss(const float* pred, const float* target, float* loss, float* g_pred,
                      int64_t n, void* stream);

/* ---- optimizer: torch.optim.Adam defaults, reference train.py:43,55 ------------- */
/* One launch over the flat fp32 parameter vector (state_dict order; the 24 tensors are
 * views of it): params, exp_avg, exp_avg_sq updated in place from grads; `step` >= 1 is
 * the 1-based step count used for the bias corrections.  No amsgrad / weight decay, as
 * in the reference.  Follow with nerf_amd_pack_weights on `params`. */
int nerf_amd_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                       int64_t n, float lr, float beta1, float beta2, float eps, int64_t step,
                       void* stream);

/* The reference's jitter draw `torch.rand(B, N)` on torch's CPU default generator
 * (utils/rendering.py:28-30), continued on the GPU with identical values: MT19937, one 32-bit
 * output per float32, u = (tempered & 0xFFFFFF) * 2^-24.  state624: the generator's 624 state
 * words (device memory); next: index of the first unread word of the current block, 0..624
 * (624 = the block is used up); out[n] receives the next n draws; state_out624 the state words
 * afterwards (equal to state624 if no new block was needed).  The host side keeps the
 * generator's counters (utils/host_rng.py reference_rand).  One workgroup; ~1 barrier per 624
 * draws; see nerf_amd_mt19937_uniform_par for long draws. */
int nerf_amd_mt19937_uniform(const uint32_t* state624, int next, float* out, int64_t n,
                             uint32_t* state_out624, void* stream);

/* The same stream produced by several workgroups: after the generator's unread words the stream is
 * cut into segments of seg_words words (a multiple of 624); segment b starts from the state
 * advanced by b * seg_words words, obtained by MT19937 jump-ahead -- polys[m][624] holds
 * x^(seg_words * 2^m) mod the characteristic polynomial (utils/mt19937_jump.npz, made and verified by
 * tools/make_mt_jump.py), m < levels, and the start states follow from the first by a doubling
 * tree of GF(2) convolutions.  levels < 0 selects the one-launch form for short draws instead:
 * polys[j-1][624] holds x^(seg_words * j), j = 1 .. -levels, and every start state is formed from the
 * first directly (up to 1 - levels segments; utils/mt19937_jump.npz holds 63 of them for 39,936-word segments).  seg_states: workspace of nerf_amd_mt19937_segments(next, n,
 * seg_words) * 624 words.  Values and final state identical to nerf_amd_mt19937_uniform. */
int64_t nerf_amd_mt19937_segments(int next, int64_t n, int64_t seg_words);
int nerf_amd_mt19937_uniform_par(const uint32_t* state624, int next, float* out, int64_t n,
                                 uint32_t* state_out624, const uint32_t* polys, int levels,
                                 int64_t seg_words, uint32_t* seg_states, void* stream);

/* The reference's range warning, utils/xyz.py:8-9 (`input not in range -1,1, check rescaling`, raised by every gamma call
 * of positional_encoder when any of the six query-point columns leaves [-1, 1]) without its two device->host syncs:
 * *word |= 1 (uint32 in DEVICE memory, an atomic OR; the caller zeroes and reads it when it likes) if any query point of
 * this call would trigger it.  rays != NULL: the points render_nerf would form from (rays[B,6], u, tbins, flags, seed,
 * ray_id0, N) exactly as the render kernels form them; only the first and the last sample of each ray are looked at (a
 * coordinate is monotone along its ray; with NERF_AMD_TS_GIVEN, whose positions need not be sorted, all N), so the cost is
 * that of 2 B points.  pts != NULL (rays NULL): B FLOATS to test -- the 6 P values of explicit points [P,6], or any
 * gamma() argument.  NaN coordinates do not warn (comparisons with NaN are false), as in the reference. */
int nerf_amd_range_check(const float* rays, const float* pts, const float* u, const float* tbins, uint32_t flags,
                         uint64_t seed, int64_t ray_id0, uint32_t* word, int64_t B, int N, void* stream);

/* ---- ray selection: RayGenerator.select + the ground-truth gather, reference utils/dataload.py:141-153, train.py:47-49 ---- */
/* The raw 32-bit outputs of the same generator (at::mt19937's random()): what `torch.randperm(n)` (dataload.py:151) draws
 * its swap positions from.  Arguments as nerf_amd_mt19937_uniform; state_out624 may be NULL. */
int nerf_amd_mt19937_raw(const uint32_t* state624, int next, uint32_t* out, int64_t n,
                         uint32_t* state_out624, void* stream);
/* HOST function (no GPU): h_poly624 = x^(624 * blocks) mod phi over GF(2), phi = the characteristic polynomial of MT19937's
 * one-word step (h_phi624: 624 little-endian words, `phi` of utils/mt19937_jump.npz).  ~0.1 s; one per table size. */
int nerf_amd_mt19937_jump_poly(int64_t blocks, const uint32_t* h_phi624, uint32_t* h_poly624);
/* state_out624 = the generator's state words (1 + q) blocks after state624's block, poly624 = x^(624 q) mod phi on the
 * device: where torch.randperm(n) leaves the generator after its n - 1 draws, of which nerf_amd_select_rays looks at the
 * first B only.  All 32 bits of all 624 words are torch's.  One launch of 16 workgroups (~85 us). */
int nerf_amd_mt19937_advance(const uint32_t* state624, const uint32_t* poly624, uint32_t* state_out624, void* stream);
/* The jitter draw that FOLLOWS the shuffle in the reference's iteration (train.py:47-51: rg.select, then render_nerf's
 * torch.rand(B, N) from the same generator) without a second dependent jump: out[n] = the n uniforms drawn from the state
 * (1 + q) blocks after state624's block, read from word next_after on -- i.e. from where nerf_amd_mt19937_advance(q) would
 * leave the generator -- with the start states of all `segments` = nerf_amd_mt19937_segments(next_after, n, seg_words) segments
 * formed from state624 in ONE launch: polys[b][624] = x^(624 (q + b * seg_words / 624)) mod phi, b < segments
 * (nerf_amd_mt19937_jump_poly).  state_out624: the state words after the draw; seg_states: workspace, segments * 624 words
 * (seg_states[0] ends up holding the state after the shuffle). */
int nerf_amd_mt19937_uniform_after(const uint32_t* state624, const uint32_t* polys, int segments, int next_after, float* out,
                                   int64_t n, uint32_t* state_out624, int64_t seg_words, uint32_t* seg_states, void* stream);
/* ray_ids = torch.randperm(n)[:B]; rays = table[ray_ids]; gt = colours[ray_ids]   (dataload.py:150-153, train.py:49) with
 * table[n,6] and colours[n,3] resident in HBM.  ids_out[B] (int64, as torch's), rays_out[B,6], gt_out[B,3]: any may be NULL.
 * The permutation prefix is the exact forward Fisher-Yates prefix of torch's CPU randperm (csrc/select.hip), from
 *   draws != NULL: draws[i], i < min(B, n-1): the generator's next 32-bit outputs (nerf_amd_mt19937_raw) -- the
 *                  reference's own ray_ids for the generator's state; seed / seed_mem ignored;
 *   draws == NULL: the counter RNG keyed by seed (+ *seed_mem if seed_mem != NULL: a uint64 in DEVICE memory read when the
 *                  kernel runs, so a launch captured into a hipGraph selects a fresh batch at every replay).
 * B <= n < 2^32 / 20 (beyond, torch.randperm is another algorithm: NERF_AMD_EUNSUP).  workspace: nerf_amd_select_workspace_bytes(B).
 * workspace 16-byte aligned.  Three launches, no atomics on global memory, deterministic. */
int64_t nerf_amd_select_workspace_bytes(int64_t B);
int nerf_amd_select_rays(const uint32_t* draws, uint64_t seed, const uint64_t* seed_mem, int64_t n, int64_t B,
                         const float* table, const float* colours, float* rays_out, float* gt_out, int64_t* ids_out,
                         void* workspace, void* stream);

/* The same update with the step-dependent scalars in DEVICE memory: hyper[6] = {lr, beta1,
 * beta2, eps, 1 - beta1^step, sqrt(1 - beta2^step)} (fp32).  The launch carries no per-step
 * argument, so it can sit inside a captured hipGraph replayed every iteration while the host
 * rewrites `hyper` (training.GraphedTrainStep). */
int nerf_amd_adam_step_hyper(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                             int64_t n, const float* hyper, void* stream);
/* The scalars of a captured step without a copy between graph launches: a ring of slots x 8 floats in PINNED HOST memory
 * (one slot per step, written by the host before it launches the step: the 6 floats of `hyper` above, then whatever the
 * caller keeps in floats 6..7 -- the training step's jitter seed offset as an int64).  nerf_amd_pinned_device_address
 * returns the address under which the device reads such a buffer (hipHostGetDevicePointer; < 0: not pinned / not
 * mapped) -- query it once, outside any capture; nerf_amd_hyper_fetch(ring_dev = that address, ...) copies slot
 * (*counter % slots) into hyper[0..7] on the device and increments *counter (device memory, uint32).  The caller reuses
 * a slot only after the step that read it has been passed by an event. */
int64_t nerf_amd_pinned_device_address(const void* host);
int nerf_amd_hyper_fetch(const float* ring_dev, int slots, float* hyper, uint32_t* counter, void* stream);

/* ---- networks of other sizes: Nerf(Lp, Ld, H), reference utils/nets.py:9-32 ---------------- */
/* The fused kernels implement the one configuration the reference constructs (Nerf() = (10, 4, 256): train.py:41,
 * test.py:27).  Every nn.Linear (+ nn.ReLU) of any other size, and its backward (dX = dY W, dW = dY^T X, db = dY^T 1),
 * is this strided fp32 GEMM on the exact-f32 MFMA:
 *     C[i*ldc + j] (+)= sum_k A(i,k) * B(k,j) (+ bias[j]) (ReLU),   i < M, j < N, k < K
 *     A(i,k) = A[i*sa_i + k*sa_k], taken as 0 where A_mask[i*sa_i + k*sa_k] <= 0 (A_mask may be NULL: the ReLU
 *              derivative of a saved activation applied to the incoming gradient);
 *     B(k,j) = B[k*sb_k + j*sb_j] (a weight matrix or a column slice of one, an activation, or one 1.0f with both
 *              strides 0 for column sums).
 * flags: NERF_AMD_LINEAR_RELU, NERF_AMD_LINEAR_ACCUMULATE (C += ; with a long K and few output tiles the reduction is
 * split and summed with float atomics: summation order then varies from run to run).  bias may be NULL. */
#define NERF_AMD_LINEAR_RELU       1u
#define NERF_AMD_LINEAR_ACCUMULATE 2u
int nerf_amd_linear_f32(const float* A, int64_t sa_i, int64_t sa_k, const float* A_mask,
                        const float* B, int64_t sb_k, int64_t sb_j, const float* bias,
                        float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, uint32_t flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_AMD_H */
