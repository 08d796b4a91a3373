// adam.hip -- the optimizer step of the training loop (reference train.py:43,55-57:
// torch.optim.Adam(lr=5e-4, betas=(0.9,0.999), eps=1e-8) + per-step exponential LR
// decay) as ONE kernel over the flat fp32 parameter vector (state_dict order),
// SURVEY.md section 8f, N3.  The 24 tensors are views of that vector on the Python
// side, so there is no per-tensor launch; the packed MFMA weight images are
// re-derived from the same vector right after (nerf_amd_pack_weights).
//
// Per element, exactly torch's single-tensor Adam (no amsgrad, no weight decay):
//   m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "nerf_device.h"

namespace {
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                            float bc1, float bc2_sqrt) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}
// The same update with its step-dependent scalars read from device memory
// (hyper = {lr, beta1, beta2, eps, 1 - beta1^t, sqrt(1 - beta2^t)}), so the launch can live
// inside a captured hipGraph that is replayed with a new learning rate / step count.
__global__ void adam_hyper_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, long long n, const float* __restrict__ hyper) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2_sqrt = hyper[5];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}
// The step's scalars from a ring in pinned HOST memory (8 floats per slot, written by the host before it launches the
// step) into the device vector the other kernels read, as a kernel: slot = *counter % slots, then *counter += 1.  A
// replayed hipGraph can so begin with its own parameters' upload -- a stream-ordered hipMemcpyAsync between two graph
// launches costs ~20 us of idle GPU at the seam (copy engine start + the launch behind it), a kernel node 2 us.
__global__ void hyper_fetch_kernel(const float* ring, int slots, float* __restrict__ hyper, unsigned* __restrict__ counter) {
    const unsigned c = *counter;
    const float v = __builtin_nontemporal_load(ring + (size_t)(c % (unsigned)slots) * 8 + threadIdx.x);
    __syncthreads();                       // every thread has read the counter
    hyper[threadIdx.x] = v;
    if (threadIdx.x == 0) *counter = c + 1;
}
}  // namespace

extern "C" int nerf_amd_launch_hyper_fetch(const float* ring_host, int slots, float* hyper, unsigned* counter, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(hyper_fetch_kernel, dim3(1), dim3(8), 0, stream, ring_host, slots, hyper, counter);
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_adam_hyper(float* params, const float* grads, float* m, float* v, long long n,
                                          const float* hyper, hipStream_t stream) {
    (void)hipGetLastError();
    if (n == 0) return 0;
    long long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adam_hyper_kernel, dim3((unsigned)grid), dim3(256), 0, stream, params, grads, m, v, n, hyper);
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_adam(float* params, const float* grads, float* m, float* v, long long n, float lr,
                                    float b1, float b2, float eps, float bc1, float bc2_sqrt, hipStream_t stream) {
    (void)hipGetLastError();
    if (n == 0) return 0;
    long long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, stream, params, grads, m, v, n, lr, b1, b2,
                       eps, bc1, bc2_sqrt);
    return (int)hipGetLastError();
}
