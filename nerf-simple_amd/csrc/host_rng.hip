// host_rng.hip -- the reference's jitter draw, utils/rendering.py:28-30
// (`u = torch.rand(B, N)` on torch's CPU default generator, then `.cuda()`), produced ON the GPU
// with the same values: torch's CPU generator is MT19937 (at::mt19937), a float32 uniform takes
// one 32-bit output, tempers it and keeps 24 bits: u = (y & 0xFFFFFF) * 2^-24
// (ATen/core/DistributionsHelper.h, uniform_real_distribution<float>).  The caller uploads the
// generator's 624 state words and its read position, this kernel continues the stream for n
// draws and hands back the state words to be written into the generator again (the 5 KB round
// trip replaces the host-side generation of B*N floats and their PCIe copy: 328 MB for an
// 800x800x128 image).  oracle/nerf_oracle.py restates the same stream in numpy.
//
// MT19937 is sequential from block to block (624 words) but wide inside a block:
//   new[i] = old[i+397] ^ T(old[i], old[i+1])            i in [0, 227)
//   new[i] = new[i-227] ^ T(old[i], old[i+1])            i in [227, 623)
//   new[623] = new[396] ^ T(old[623], new[0])
// Producer thread t < 227 (waves 0..3) owns words t, t+227 and t+454 of every block and keeps them
// in registers: its chain t -> t+227 -> t+454 needs only its own previous results, its own words
// of the old block and four neighbours' words (old[t+1], old[t+228], old[t+455], old[t+397]),
// which travel through a double-buffered LDS copy -- one barrier and one LDS round trip per
// block.  Word 623 is formed by every thread from two broadcast reads and carried in a register.
// Waves 4..7 are consumers: while the producers form block b+1 they read block b from LDS,
// temper, convert and store it (coalesced).  One workgroup (the stream is one sequence).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int MT_N = 624, MT_M = 397, MT_D = MT_N - MT_M;   // 227

__device__ __forceinline__ unsigned mt_twist(unsigned u, unsigned v) {
    const unsigned y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ float mt_uniform(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return (float)(y & 0xffffffu) * 5.9604644775390625e-08f;   // 2^-24, exact
}

__global__ __launch_bounds__(512) void mt19937_uniform_kernel(const unsigned* __restrict__ state_in, int next0,
                                                              float* __restrict__ out, long long n,
                                                              unsigned* __restrict__ state_out) {
    __shared__ unsigned buf[2][MT_N];
    const int t = threadIdx.x;
    const bool producer = t < 256;
    const bool own = t < MT_D;                         // producer that owns words t, t+227 and (t < 169) t+454
    const bool own3 = t + 2 * MT_D < MT_N - 1;
    const int c = t - 256;                             // consumer index 0..255
    for (int i = t; i < MT_N; i += 512) buf[0][i] = state_in[i];
    __syncthreads();
    unsigned r0 = 0, r1 = 0, r2 = 0;
    if (own) {
        r0 = buf[0][t];
        r1 = buf[0][t + MT_D];
        if (own3) r2 = buf[0][t + 2 * MT_D];
    }
    unsigned last = buf[0][MT_N - 1];                   // word 623 of the current block
    int cur = 0, lo = next0;                           // words [lo, 624) of the current block are unread
    long long pos = 0;                                 // draws written so far
    while (true) {
        const unsigned* old = buf[cur];
        const bool more = pos + (MT_N - lo) < n;       // uniform: another block will be needed
        if (producer) {
            if (more && own) {
                // the next block from this one: own words in registers, four neighbours' words from LDS
                const unsigned a1 = old[t + 1];
                const unsigned b1 = old[t + MT_D + 1];
                const unsigned m = (t + MT_M == MT_N - 1) ? last : old[t + MT_M];
                unsigned c1 = 0;
                if (own3) c1 = (t + 2 * MT_D + 1 == MT_N - 1) ? last : old[t + 2 * MT_D + 1];
                unsigned* nw = buf[cur ^ 1];
                r0 = m ^ mt_twist(r0, a1);
                r1 = r0 ^ mt_twist(r1, b1);
                nw[t] = r0;
                nw[t + MT_D] = r1;
                if (own3) {
                    r2 = r1 ^ mt_twist(r2, c1);
                    nw[t + 2 * MT_D] = r2;
                }
            }
        } else {
            // this block's unread words [lo, 624) -> out[pos + (idx - lo)]
            float* o = out + (pos - lo);
            const long long end = n - (pos - lo);      // idx must stay below this
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = c + 256 * k;
                if (idx < MT_N && idx >= lo && idx < end) o[idx] = mt_uniform(idx == MT_N - 1 ? last : old[idx]);
            }
        }
        pos += MT_N - lo;
        lo = 0;
        if (!more) break;
        // LDS-only barrier: __syncthreads() would also wait (vmcnt(0)) for the output stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur ^= 1;
        last = buf[cur][MT_M - 1] ^ mt_twist(last, buf[cur][0]);
    }
    // the state words afterwards: the producers' registers when blocks were formed, else the input
    if (own) {
        state_out[t] = r0;
        state_out[t + MT_D] = r1;
        if (own3) state_out[t + 2 * MT_D] = r2;
    }
    if (t == 255) state_out[MT_N - 1] = last;
}

}  // namespace

extern "C" int nerf_amd_launch_mt19937_uniform(const uint32_t* state_in, int next, float* out, long long n,
                                               uint32_t* state_out, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(mt19937_uniform_kernel, dim3(1), dim3(512), 0, stream, state_in, next, out, n, state_out);
    return (int)hipGetLastError();
}
