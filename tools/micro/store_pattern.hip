// Compute-free replay of the training kernels' activation-store pattern:
// 256 persistent workgroups of 8 waves walk 256-point tiles; per tile and layer every wave issues
// 8 fragments x 2 column blocks of buffer_store_dwordx4 into the point-blocked block
// [feature/8 (32)][point (256)][16 B] -- 16 lanes x 16 B = 256 contiguous bytes per quarter-wave,
// four such runs 4 KiB apart per instruction.  Variant 1 writes each wave's 1 KiB contiguously
// instead (an upper bound for any layout).  Prints the sustained write bandwidth.
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr long long BLOCK = 256 * 512;   // one (layer, tile) block

template <int VARIANT>
__global__ __launch_bounds__(512) void store_kernel(char* base, long long ntiles, int layers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int chunk = (g & 1) * 2 + (g >> 1);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int L = 0; L < layers; ++L) {
            char* tb = base + ((long long)L * ntiles + tile) * BLOCK;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tb, 0, (int)BLOCK, 0x00020000);
#pragma unroll
            for (int Q = 0; Q < 8; ++Q)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int local = wave * 32 + cb * 16 + col;
                    int off;
                    if (VARIANT == 0) off = chunk * 4096 + local * 16 + Q * 16384;
                    else off = ((Q * 2 + cb) * 8 + wave) * 1024 + lane * 16;          // 1 KiB contiguous per wave-instruction
                    const u32x4 v = {(unsigned)tile, (unsigned)L, (unsigned)Q, (unsigned)lane};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
                }
        }
    }
}

int main() {
    const long long P = 262144 * 4, ntiles = P / 256;          // 4 x the 4096 x 64 training batch: 5.4 GB
    const int layers = 10;
    char* buf;
    if (hipMalloc(&buf, (size_t)layers * ntiles * BLOCK) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            if (variant == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(256), dim3(512), 0, 0, buf, ntiles, layers);
            else hipLaunchKernelGGL(store_kernel<1>, dim3(256), dim3(512), 0, 0, buf, ntiles, layers);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = (double)layers * ntiles * BLOCK;
        printf("variant %d (%s): %.3f ms for %.2f GB -> %.2f TB/s\n", variant,
               variant == 0 ? "training kernels' pattern" : "1 KiB contiguous per wave-instruction", best, bytes / 1e9,
               bytes / best / 1e9);
    }
    return 0;
}
