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

// the three images of a training step in ONE launch: forward bf16 image, its bias table, backward image.  The status
// block is cleared here, and the weight-range word is DECIDED here, for exactly the weights packed now: a clear by one
// workgroup and a set by another would land in no defined order (two XCDs' L2s writing one word), so the workgroups
// collect their findings with device-scope atomics in two scratch words of the block and the last one to finish writes
// the verdict (and leaves the scratch words zero for the next launch).  Weights repaired through the flat vector therefore
// stop being flagged at the next re-pack.
constexpr int PACK_SCRATCH_FLAG = 8, PACK_SCRATCH_COUNT = 9;       // words of the status block (with 16..23), zero between launches
// One 1 KiB fragment (512 elements, two per thread) at a time per workgroup: which image, layer, row tile and k-step a
// fragment is follows from its number alone -- scalar code, once per fragment -- and a thread is left with a few bit
// operations and one multiply-add per element.  One workgroup per fragment (2289); their arrivals at the verdict
// protocol below are counted on 8 sharded words first (one word takes ~90 atomic arrivals per microsecond:
// MI355X_MICROARCH.md), the last arriver of each shard on a second word.
constexpr int PACK_TRAIN_BIAS_FRAGS = (F32_BIAS_FLOATS + 511) / 512;
constexpr int PACK_TRAIN_FRAGS = B16_WEIGHT_KIB + BWD_WEIGHT_KIB + PACK_TRAIN_BIAS_FRAGS;
constexpr int PACK_TRAIN_WGS = PACK_TRAIN_FRAGS;         // one fragment each: the gathers of a fragment are latency, not bandwidth
constexpr int PACK_SHARDS = 8, PACK_SCRATCH_SHARD0 = 16; // arrival counters: one per shard (words 16..23), then word 9
__host__ __device__ constexpr unsigned pack_shard_size(unsigned shard, unsigned wgs) { return (wgs - shard + PACK_SHARDS - 1) / PACK_SHARDS; }
__global__ __launch_bounds__(256) void pack_train_kernel(const float* __restrict__ params, __bf16* __restrict__ img,
                                                         float* __restrict__ bias, __bf16* __restrict__ bwd,
                                                         unsigned* __restrict__ status) {
    if (blockIdx.x == 0 && threadIdx.x < B16_STATUS_BYTES / 4 && threadIdx.x != NERF_STATUS_WORD_WEIGHT_RANGE &&
        threadIdx.x != PACK_SCRATCH_FLAG && threadIdx.x != PACK_SCRATCH_COUNT &&
        !(threadIdx.x >= PACK_SCRATCH_SHARD0 && threadIdx.x < PACK_SCRATCH_SHARD0 + PACK_SHARDS))
        status[threadIdx.x] = 0u;
    bool bad = false;
    for (int frag = blockIdx.x; frag < PACK_TRAIN_FRAGS; frag += gridDim.x) {
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
                bad |= !(__builtin_fabsf((float)cv) < __builtin_inff());                                  // as pack_b16_kernel
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
    }
    // The verdict.  Only the two scratch words travel between workgroups, and they travel by device-scope atomics (executed
    // at the memory side, coherent by themselves): no fence -- a __threadfence() here is an L2 write-back per workgroup and
    // cost 45 us of a 60 us kernel.  A workgroup's OR has been performed before its arrival is counted (the OR returns its
    // old value, which is waited for), so the workgroup that draws the last ticket reads the OR of all.
    if (__syncthreads_or(bad) && threadIdx.x == 0) {
        const unsigned before = atomicOr(&status[PACK_SCRATCH_FLAG], 1u);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(before) : "memory");
    }
    if (threadIdx.x == 0) {
        unsigned* shard = &status[PACK_SCRATCH_SHARD0 + (blockIdx.x % PACK_SHARDS)];
        if (atomicAdd(shard, 1u) == pack_shard_size(blockIdx.x % PACK_SHARDS, gridDim.x) - 1) {          // last of its shard
            atomicExch(shard, 0u);
            if (atomicAdd(&status[PACK_SCRATCH_COUNT], 1u) == PACK_SHARDS - 1) {       // last shard: every workgroup has reported
                atomicExch(&status[NERF_STATUS_WORD_WEIGHT_RANGE], atomicExch(&status[PACK_SCRATCH_FLAG], 0u));
                atomicExch(&status[PACK_SCRATCH_COUNT], 0u);
            }
        }
    }
}

}  // namespace

extern "C" int nerf_amd_launch_pack_train(const float* params, void* packed_bf16, void* packed_bwd, hipStream_t stream) {
    (void)hipGetLastError();
    char* img = reinterpret_cast<char*>(packed_bf16);
    hipLaunchKernelGGL(pack_train_kernel, dim3(PACK_TRAIN_WGS), dim3(256), 0, stream, params, reinterpret_cast<__bf16*>(img),
                       reinterpret_cast<float*>(img + (long long)B16_WEIGHT_KIB * 1024), reinterpret_cast<__bf16*>(packed_bwd),
                       reinterpret_cast<unsigned*>(img + B16_STATUS_OFF));
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
