// raygen.hip -- pinhole ray generation on the device:
// the ray table the reference builds on the CPU with rays_single_cam + R @ dirs
// (utils/xyz.py:38-52, utils/rendering.py:129-134, utils/dataload.py:114-129).
//
//   pixel p = h*W + w (row-major, no half-pixel offset)
//   dir_cam(h,w) = ((w - W//2)/f, -(h - H//2)/f, -1)
//   rays[p] = [pose[:3,3], pose[:3,:3] @ dir_cam]
//
// HBM-bound and tiny: 24 B written per ray, nothing read.
#include "nerf_device.h"

namespace {

struct Pose { float r[9]; float t[3]; };

__global__ void generate_rays_kernel(Pose pose, int H, int W, float f, long long ray0, long long n,
                                     float* __restrict__ rays) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const long long p = ray0 + i;
        const int h = (int)(p / W), w = (int)(p - (long long)h * W);
        const float dx = __fdiv_rn((float)(w - W / 2), f);
        const float dy = -__fdiv_rn((float)(h - H / 2), f);
        const float dz = -1.0f;
        float* o = rays + i * 6;
        o[0] = pose.t[0]; o[1] = pose.t[1]; o[2] = pose.t[2];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            o[3 + k] = __fmaf_rn(pose.r[3 * k + 2], dz, __fmaf_rn(pose.r[3 * k + 1], dy, __fmul_rn(pose.r[3 * k], dx)));
    }
}

}  // namespace

extern "C" int nerf_amd_launch_generate_rays(const float* h_pose, int H, int W, float f, long long ray0,
                                             long long n, float* rays, hipStream_t stream) {
    (void)hipGetLastError();
    if (n == 0) return 0;
    Pose p;
    // h_pose: row-major 3x4 (or the top of a 4x4): [R | t]
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) p.r[3 * r + c] = h_pose[4 * r + c];
        p.t[r] = h_pose[4 * r + 3];
    }
    long long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(generate_rays_kernel, dim3((unsigned)g), dim3(256), 0, stream, p, H, W, f, ray0, n, rays);
    return (int)hipGetLastError();
}
