// pack.hip -- fp32 state dict (flat, reference state_dict order) -> the
// MFMA-fragment-ordered weight images streamed by the fused kernels.
// Runs once per parameter update; one thread per packed element.
#include "nerf_device.h"

using namespace nerf_layout;

namespace {

// 16-bit image (bf16 or fp16), 16-row tiles: [tile (layer, rt)][k-step][lane][8 elements]
template <class T>
__global__ void pack_b16_kernel(const float* __restrict__ params, T* __restrict__ out, unsigned* __restrict__ status) {
    const long long total = (long long)B16_WEIGHT_KIB * 512;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int kib = (int)(e >> 9);
        int L = 0;
        while (L + 1 < NUM_LAYERS && kib >= b16_layer_off_kib(L + 1)) ++L;
        const int rel = kib - b16_layer_off_kib(L);
        const int rt = rel / b16_ks(L), s = rel % b16_ks(L);
        const int lane = (int)(e >> 3) & 63, j = (int)e & 7;
        const int row = 16 * rt + (lane & 15), g = lane >> 4;
        const float wv = weight_at(params, L, row, src_col_b16(L, s, g, j));
        const T cv = (T)wv;
        out[e] = cv;
        // a weight that is not finite in the operand type -- beyond its range (fp16: |w| > 65504), or NaN / inf to
        // begin with (a diverged run): sticky flag (the status block was zeroed by pack_bias_kernel, launched in front
        // of this kernel)
        if (!(__builtin_fabsf((float)cv) < __builtin_inff())) status[NERF_STATUS_WORD_WEIGHT_RANGE] = 1u;
    }
}

// backward image: [tile (bwd layer b, rt)][k-step][lane][8 bf16] = W_wl[out(ks,g,j)][in = 16rt + r]
__global__ void pack_bwd_kernel(const float* __restrict__ params, __bf16* __restrict__ out) {
    const long long total = (long long)BWD_WEIGHT_KIB * 512;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int kib = (int)(e >> 9);
        int b = 0;
        while (b + 1 < NUM_BWD && kib >= bwd_layer_off_kib(b + 1)) ++b;
        const int rel = kib - bwd_layer_off_kib(b);
        const int rt = rel / bwd_ks(b), s = rel % bwd_ks(b);
        const int lane = (int)(e >> 3) & 63, j = (int)e & 7;
        const int in_col = 16 * rt + (lane & 15), g = lane >> 4;
        const int o = bwd_src_out(b, s, g, j);
        out[e] = (__bf16)(o < 0 ? 0.f : weight_at(params, bwd_desc(b).wl, o, in_col));
    }
}

// f32 image: [chunk (layer, t)][k-step/4][lane][4 f32]
__global__ void pack_f32_kernel(const float* __restrict__ params, float* __restrict__ out) {
    const long long total = (long long)F32_WEIGHT_KIB * 256;      // floats
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int kib = (int)(e >> 8);
        int L = 0;
        while (L + 1 < NUM_LAYERS && kib >= f32_layer_off_kib(L + 1)) ++L;
        const int rel = kib - f32_layer_off_kib(L);
        const int ck = f32_chunk_kib(L);
        const int t = rel / ck, q = rel % ck;           // q: group of 4 k-steps (1 KiB)
        const int lane = (int)(e >> 2) & 63, i = (int)e & 3;
        const int ks = 4 * q + i;
        float v = 0.f;
        if (ks < f32_ks(L)) {
            const int row = 16 * t + (lane & 15), g = lane >> 4;
            v = weight_at(params, L, row, src_col_f32(L, ks, g));
        }
        out[e] = v;
    }
}

// bias table: natural row order, 16 rows per tile (the 16-bit and the f32 kernels share it)
// status != NULL (16-bit images): also zero the status block behind the bias table
__global__ void pack_bias_kernel(const float* __restrict__ params, float* __restrict__ out, unsigned* __restrict__ status) {
    static_assert(B16_BIAS_FLOATS == F32_BIAS_FLOATS, "one bias table layout");
    if (status && blockIdx.x == 0 && threadIdx.x < B16_STATUS_BYTES / 4) status[threadIdx.x] = 0u;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < F32_BIAS_FLOATS; e += gridDim.x * blockDim.x) {
        int L = 0;
        while (L + 1 < NUM_LAYERS && e >= f32_bias_off(L + 1)) ++L;
        out[e] = bias_at(params, L, e - f32_bias_off(L));
    }
}

// the three images of a training step in ONE launch: forward bf16 image, its bias table, backward image.  One 1 KiB
// fragment (512 elements, two per thread) per workgroup: which image, layer, row tile and k-step a fragment is follows from
// its number alone -- scalar code, once per workgroup -- and a thread is left with a few bit operations and one
// multiply-add per element.
// The status block, weight-range word included, is cleared by pack_clear_status_kernel, launched in front (a kernel, not a
// memset node: csrc/dw_gemm.hip has the story; kernel -> kernel edges order correctly in a captured graph), so the word
// states the verdict for exactly the weights packed now: weights repaired through the flat vector stop being flagged at the
// next re-pack.  (A clear and a set from two workgroups of ONE kernel reach memory through two XCDs' L2s in no defined
// order; deciding the word inside the kernel by a last-arriver protocol -- device-scope atomics on sharded counters --
// was built and measured: 34 us against 16 for this kernel, the 2289 returning atomics keep every workgroup alive for a
// memory round trip.  The extra launch costs 2-3 us.)
__global__ void pack_clear_status_kernel(unsigned* __restrict__ status) {
    if (threadIdx.x < B16_STATUS_BYTES / 4) status[threadIdx.x] = 0u;
}
constexpr int PACK_TRAIN_BIAS_FRAGS = (F32_BIAS_FLOATS + 511) / 512;
constexpr int PACK_TRAIN_FRAGS = B16_WEIGHT_KIB + BWD_WEIGHT_KIB + PACK_TRAIN_BIAS_FRAGS;
__global__ __launch_bounds__(256) void pack_train_kernel(const float* __restrict__ params, __bf16* __restrict__ img,
                                                         float* __restrict__ bias, __bf16* __restrict__ bwd,
                                                         unsigned* __restrict__ status) {
    bool bad = false;
    const int frag = blockIdx.x;
    if (frag < B16_WEIGHT_KIB) {
        const int kib = frag;
        int L = 0;
        while (L + 1 < NUM_LAYERS && kib >= b16_layer_off_kib(L + 1)) ++L;
        const int rel = kib - b16_layer_off_kib(L);
        const int rt = rel / b16_ks(L), s_ = rel % b16_ks(L);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = threadIdx.x + 256 * h;                 // element of the fragment: [lane][8]
            const int lane = e >> 3, j = e & 7;
            const __bf16 cv = (__bf16)weight_at(params, L, 16 * rt + (lane & 15), src_col_b16(L, s_, lane >> 4, j));
            img[(long long)kib * 512 + e] = cv;
            bad |= !(__builtin_fabsf((float)cv) < __builtin_inff());                                      // as pack_b16_kernel
        }
    } else if (frag < B16_WEIGHT_KIB + BWD_WEIGHT_KIB) {
        const int kib = frag - B16_WEIGHT_KIB;
        int b = 0;
        while (b + 1 < NUM_BWD && kib >= bwd_layer_off_kib(b + 1)) ++b;
        const int rel = kib - bwd_layer_off_kib(b);
        const int rt = rel / bwd_ks(b), s_ = rel % bwd_ks(b);
        const int wl = bwd_desc(b).wl;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = threadIdx.x + 256 * h;
            const int lane = e >> 3, j = e & 7;
            const int o = bwd_src_out(b, s_, lane >> 4, j);
            bwd[(long long)kib * 512 + e] = (__bf16)(o < 0 ? 0.f : weight_at(params, wl, o, 16 * rt + (lane & 15)));
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = (frag - B16_WEIGHT_KIB - BWD_WEIGHT_KIB) * 512 + threadIdx.x + 256 * h;
            if (q < F32_BIAS_FLOATS) {
                int L = 0;
                while (L + 1 < NUM_LAYERS && q >= f32_bias_off(L + 1)) ++L;
                bias[q] = bias_at(params, L, q - f32_bias_off(L));
            }
        }
    }
    if (bad) status[NERF_STATUS_WORD_WEIGHT_RANGE] = 1u;          // sticky within the launch, cleared in front of it
}

}  // namespace

extern "C" int nerf_amd_launch_pack_train(const float* params, void* packed_bf16, void* packed_bwd, hipStream_t stream) {
    (void)hipGetLastError();
    char* img = reinterpret_cast<char*>(packed_bf16);
    unsigned* status = reinterpret_cast<unsigned*>(img + B16_STATUS_OFF);
    hipLaunchKernelGGL(pack_clear_status_kernel, dim3(1), dim3(64), 0, stream, status);
    hipLaunchKernelGGL(pack_train_kernel, dim3(PACK_TRAIN_FRAGS), dim3(256), 0, stream, params, reinterpret_cast<__bf16*>(img),
                       reinterpret_cast<float*>(img + (long long)B16_WEIGHT_KIB * 1024), reinterpret_cast<__bf16*>(packed_bwd), status);
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_pack(const float* params, void* packed, int precision, hipStream_t stream) {
    (void)hipGetLastError();   // drop any stale error: the return value is about THIS launch
    char* img = reinterpret_cast<char*>(packed);
    if (precision == 1 || precision == 2) {
        unsigned* status = reinterpret_cast<unsigned*>(img + B16_STATUS_OFF);
        hipLaunchKernelGGL(pack_bias_kernel, dim3(10), dim3(256), 0, stream, params,
                           reinterpret_cast<float*>(img + (long long)B16_WEIGHT_KIB * 1024), status);
        if (precision == 1)
            hipLaunchKernelGGL(pack_b16_kernel<__bf16>, dim3(1024), dim3(256), 0, stream, params,
                               reinterpret_cast<__bf16*>(img), status);
        else
            hipLaunchKernelGGL(pack_b16_kernel<_Float16>, dim3(1024), dim3(256), 0, stream, params,
                               reinterpret_cast<_Float16*>(img), status);
    } else if (precision == 3) {
        // training backward image (bf16)
        hipLaunchKernelGGL(pack_bwd_kernel, dim3(1024), dim3(256), 0, stream, params,
                           reinterpret_cast<__bf16*>(img));
    } else {
        hipLaunchKernelGGL(pack_f32_kernel, dim3(1024), dim3(256), 0, stream, params,
                           reinterpret_cast<float*>(img));
        hipLaunchKernelGGL(pack_bias_kernel, dim3(10), dim3(256), 0, stream, params,
                           reinterpret_cast<float*>(img + (long long)F32_WEIGHT_KIB * 1024), static_cast<unsigned*>(nullptr));
    }
    return (int)hipGetLastError();
}
