// Does a buffer written by one kernel come back faster when the next kernel reads it at once?  ("what
// comes next": dY could travel from the dX chain to the dW products without touching HBM only if the 256 MB memory-side
// cache keeps what was just written.)  For footprints from 32 MB to 1 GB: a streaming write of the buffer (plain or
// non-temporal stores), then a streaming read of it (plain or non-temporal loads), the read timed alone; "cold" = the same
// read after 2 GB of unrelated traffic.  Prints the read bandwidth per footprint and policy.
//   hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip && ./mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void write_kernel(u32x4* p, long long n, unsigned tag) {
    const u32x4 v = {tag, tag + 1, tag + 2, tag + 3};
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        if (NT) __builtin_nontemporal_store(v, p + i);
        else p[i] = v;
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* p, long long n, unsigned* sink, bool reverse) {
    u32x4 acc = {0, 0, 0, 0};
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        const long long j = reverse ? n - 1 - i : i;
        acc ^= NT ? __builtin_nontemporal_load(p + j) : p[j];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main() {
    const size_t MB = 1 << 20;
    char *buf, *other;
    unsigned* sink;
    if (hipMalloc(&buf, 1024 * MB) != hipSuccess || hipMalloc(&other, 2048 * MB) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * 8;
    printf("footprint MB | read GB/s: plain write+plain read | NT write+NT read | plain write, read in reverse order | cold (after 2 GB of other traffic)\n");
    for (size_t mb : {32, 64, 128, 192, 256, 384, 512, 1024}) {
        const long long n = (long long)(mb * MB / 16);
        float res[4] = {0, 0, 0, 0};
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                if (mode == 1) hipLaunchKernelGGL(write_kernel<true>, dim3(grid), dim3(256), 0, 0, (u32x4*)buf, n, (unsigned)rep);
                else hipLaunchKernelGGL(write_kernel<false>, dim3(grid), dim3(256), 0, 0, (u32x4*)buf, n, (unsigned)rep);
                if (mode == 3) hipLaunchKernelGGL(write_kernel<false>, dim3(grid), dim3(256), 0, 0, (u32x4*)other, (long long)(2048 * MB / 16), 7u);
                (void)hipEventRecord(e0);
                if (mode == 1) hipLaunchKernelGGL(read_kernel<true>, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, n, sink, false);
                else hipLaunchKernelGGL(read_kernel<false>, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, n, sink, mode == 2);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            res[mode] = (float)(mb * MB) / best / 1e6f;
        }
        printf("%9zu | %8.0f | %8.0f | %8.0f | %8.0f\n", mb, res[0], res[1], res[2], res[3]);
    }
    return 0;
}
