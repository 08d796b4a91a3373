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
// temper, convert and store it (coalesced).  A workgroup produces one sequential piece of the
// stream; long draws are cut into segments whose start states come from jump-ahead (below).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int MT_N = 624, MT_M = 397, MT_D = MT_N - MT_M;   // 227

__device__ __forceinline__ unsigned mt_twist(unsigned u, unsigned v) {
    const unsigned y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
// one draw as the consumer wants it: a float32 uniform (torch.rand) or the 32-bit output itself (at::mt19937's
// random(), what torch.randperm takes its swap positions from)
template <typename T> __device__ __forceinline__ T mt_draw(unsigned y);
template <> __device__ __forceinline__ float mt_draw<float>(unsigned y) {
    return (float)(mt_temper(y) & 0xffffffu) * 5.9604644775390625e-08f;   // 2^-24, exact
}
template <> __device__ __forceinline__ unsigned mt_draw<unsigned>(unsigned y) { return mt_temper(y); }

// Segment b of the stream (blockIdx.x): b = 0 starts from the generator's state (unread words from
// next0, then seg_words draws of new blocks); b >= 1 starts from seg_states[b] = that state advanced by
// b * seg_words words (mt19937_jump_kernel), all of it unread-exhausted, and produces seg_words draws.
// The last segment also hands back the state words.  A single segment (gridDim.x == 1) is the whole call.
template <typename T>
__global__ __launch_bounds__(512) void mt19937_uniform_kernel(const unsigned* __restrict__ state_in0, int next00,
                                                              T* __restrict__ out0, long long n_total,
                                                              unsigned* __restrict__ state_out,
                                                              const unsigned* __restrict__ seg_states,
                                                              long long seg_words) {
    __shared__ unsigned buf[2][MT_N];
    const int t = threadIdx.x;
    const int seg = blockIdx.x;
    const bool last_seg = seg == (int)gridDim.x - 1;
    const long long avail0 = MT_N - next00;                         // unread words of the generator's block
    const unsigned* state_in = seg == 0 ? state_in0 : seg_states + (long long)seg * MT_N;
    const int next0 = seg == 0 ? next00 : MT_N;
    const long long off = seg == 0 ? 0 : avail0 + (long long)seg * seg_words;
    T* out = out0 + off;
    long long n = n_total - off;
    if (gridDim.x > 1) {
        const long long cap = seg == 0 ? avail0 + seg_words : seg_words;
        if (n > cap) n = cap;
    }
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
            T* o = out + (pos - lo);
            const long long end = n - (pos - lo);      // idx must stay below this
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = c + 256 * k;
                if (idx < MT_N && idx >= lo && idx < end) o[idx] = mt_draw<T>(idx == MT_N - 1 ? last : old[idx]);
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
    if (!last_seg || !state_out) return;
    // the state words afterwards: the producers' registers when blocks were formed, else the input
    if (own) {
        state_out[t] = r0;
        state_out[t + MT_D] = r1;
        if (own3) state_out[t + 2 * MT_D] = r2;
    }
    if (t == 255) state_out[MT_N - 1] = last;
}

// ---- jump-ahead -------------------------------------------------------------------------------
// states[dst0 + j] = F^J states[src0 + j], j = blockIdx.x, F = MT19937's one-word step, J given by
// poly = x^J mod phi(x) (tools/make_mt_jump.py; 19937 bits in 624 words).  With w_0, w_1, ... the raw
// word sequence that starts with the source state's 624 words, F^i s is the window w_i .. w_{i+623},
// so (F^J s)[k] = XOR over the set bits i of poly of w[i + k]: a GF(2) convolution.  The workgroup
// first extends the sequence to 33 blocks in LDS (83 KiB with the polynomial; 32 block steps with the chain scheme of
// the generator), then each thread accumulates its three output words -- the bit test is
// wave-uniform, the LDS reads of consecutive threads consecutive.  The 31 low bits of word 0 of a
// state are not part of it (a block-aligned state is only ever regenerated from): they come out
// arbitrary and are never read.
constexpr int MT_JUMP_BLOCKS = 33;                       // 33 * 624 = 20592 >= 19937 + 623 words
constexpr int MT_DEG = 19937;

constexpr int MT_JUMP_LDS_MAX = (MT_JUMP_BLOCKS + 2) * MT_N * (int)sizeof(unsigned);   // skip = 1: one more block
constexpr int MT_JUMP_SPLIT = 16;                        // workgroups sharing one jump's convolution (624 = 16 * 39 words)

// Jump j of a launch reads state src0 + j * src_step with polynomial polys[j * poly_step]: the doubling tree uses
// (src_step, poly_step) = (1, 0) -- 2^m states advanced by the same distance -- and the one-launch form (0, 1): every
// start state straight from state 0, each with its own polynomial.
// skip = 1 convolves the sequence one block later: states[dst] = F^J (the block AFTER the source state).  Every word
// the convolution then reads is a generated one, so all 32 bits of all 624 output words are the generator's own --
// which a state that is handed back to torch's generator needs (nerf_amd_mt19937_advance); with skip = 0 the 31 low bits
// of output word 0 depend on bits of the source's word 0 that are not part of its state (see above).
__global__ __launch_bounds__(256) void mt19937_jump_kernel(unsigned* __restrict__ states, const unsigned* __restrict__ polys,
                                                           int src0, int dst0, int src_step, int poly_step, int skip,
                                                           const unsigned* __restrict__ src_ext) {
    // blockIdx.x = jump * MT_JUMP_SPLIT + part: the parts of a jump each rebuild the word sequence (cheap)
    // and convolve a slice of the polynomial, combining into the zero-initialised destination with atomicXor
    const int jump = blockIdx.x / MT_JUMP_SPLIT, part = blockIdx.x % MT_JUMP_SPLIT;
    extern __shared__ unsigned w[];                       // [(MT_JUMP_BLOCKS + skip) * 624] + the polynomial [624]
    unsigned* gp = w + (MT_JUMP_BLOCKS + skip) * MT_N;    // (624 dependent global loads cost 0.6 ms)
    const int t = threadIdx.x;
    const unsigned* src = src_ext ? src_ext : states + (long long)(src0 + jump * src_step) * MT_N;
    const unsigned* poly = polys + (long long)jump * poly_step * MT_N;
    for (int i = t; i < MT_N; i += 256) {
        w[i] = src[i];
        gp[i] = poly[i];
    }
    __syncthreads();
    const bool own = t < MT_D, own3 = t + 2 * MT_D < MT_N - 1;
    for (int b = 0; b < MT_JUMP_BLOCKS - 1 + skip; ++b) {
        const unsigned* old = w + b * MT_N;
        unsigned* nw = w + (b + 1) * MT_N;
        if (own) {
            const unsigned x = old[t + MT_M] ^ mt_twist(old[t], old[t + 1]);
            const unsigned y = x ^ mt_twist(old[t + MT_D], old[t + MT_D + 1]);
            nw[t] = x;
            nw[t + MT_D] = y;
            if (own3) nw[t + 2 * MT_D] = y ^ mt_twist(old[t + 2 * MT_D], old[t + 2 * MT_D + 1]);
        }
        __syncthreads();
        if (t == 0) nw[MT_N - 1] = nw[MT_M - 1] ^ mt_twist(old[MT_N - 1], nw[0]);
        __syncthreads();
    }
    // Branch-free: every bit position is visited and the loaded words are masked with the (scalar,
    // wave-uniform) polynomial bit, so the LDS reads of 8 positions are in flight together.  A loop
    // over the set bits only (half as many reads) exposed a full LDS round trip per bit: 0.83 ms.
    unsigned a0 = 0, a1 = 0, a2 = 0;
    const bool third = t + 512 < MT_N;
    const int t2 = third ? t + 512 : t;                   // threads without a third word re-read their first
    constexpr int PER = MT_N / MT_JUMP_SPLIT;
    static_assert(PER * MT_JUMP_SPLIT == MT_N, "polynomial words per part");
    for (int iw = part * PER; iw < (part + 1) * PER; ++iw) {
        const unsigned g = __builtin_amdgcn_readfirstlane(gp[iw]);
        if (g == 0) continue;
        const unsigned* wi = w + skip * MT_N + iw * 32 + t;
#pragma unroll 8
        for (int b = 0; b < 32; ++b) {
            const unsigned m = 0u - ((g >> b) & 1u);
            a0 ^= wi[b] & m;
            a1 ^= wi[b + 256] & m;
            a2 ^= wi[b + (t2 - t)] & m;
        }
    }
    if (!third) a2 = 0;
    unsigned* dst = states + (long long)(dst0 + jump) * MT_N;
    atomicXor(dst + t, a0);
    atomicXor(dst + t + 256, a1);
    if (third) atomicXor(dst + t + 512, a2);
}

__global__ void mt19937_zero_kernel(unsigned* __restrict__ words) { words[blockIdx.x * blockDim.x + threadIdx.x] = 0u; }

}  // namespace

// Parallel form: the stream is cut into segments of seg_words words (a multiple of 624) after the
// generator's unread words; polys[m] = x^(seg_words * 2^m) mod phi; seg_states: workspace [S][624].
// levels > 0: polys[m] = x^(seg_words * 2^m) mod phi, doubling tree (up to 2^levels segments);
// levels < 0: polys[j-1] = x^(seg_words * j) mod phi, j = 1 .. -levels, every start state in ONE launch.
extern "C" int nerf_amd_launch_mt19937_uniform_par(const uint32_t* state_in, int next, float* out, long long n,
                                                   uint32_t* state_out, const uint32_t* polys, int levels,
                                                   long long seg_words, uint32_t* seg_states, hipStream_t stream) {
    (void)hipGetLastError();
    const long long avail = MT_N - next;
    long long S = 1;
    if (n > avail + seg_words) S = 1 + (n - avail - seg_words + seg_words - 1) / seg_words;
    if (levels >= 0 ? S > (1ll << levels) : S > 1 - (long long)levels) return -2;
    if (S > 1) {
        hipError_t e = hipMemsetAsync(seg_states, 0, (size_t)S * MT_N * sizeof(uint32_t), stream);   // atomicXor targets
        if (e != hipSuccess) return (int)e;
        e = hipMemcpyAsync(seg_states, state_in, MT_N * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return (int)e;
        const int lds = (MT_JUMP_BLOCKS + 1) * MT_N * (int)sizeof(unsigned);
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(mt19937_jump_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, MT_JUMP_LDS_MAX);
        if (e != hipSuccess) return (int)e;
        if (levels < 0) {                                           // every start state from state 0, one launch
            hipLaunchKernelGGL(mt19937_jump_kernel, dim3((unsigned)((S - 1) * MT_JUMP_SPLIT)), dim3(256), lds, stream, seg_states,
                               polys, 0, 1, 0, 1, 0, (const unsigned*)nullptr);
        } else {
            for (int m = 0; (1ll << m) < S; ++m) {                  // doubling tree over the segment start states
                const long long have = 1ll << m;
                const long long count = S - have < have ? S - have : have;
                hipLaunchKernelGGL(mt19937_jump_kernel, dim3((unsigned)(count * MT_JUMP_SPLIT)), dim3(256), lds, stream,
                                   seg_states, polys + (long long)m * MT_N, 0, (int)have, 1, 0, 0, (const unsigned*)nullptr);
            }
        }
    }
    hipLaunchKernelGGL(mt19937_uniform_kernel<float>, dim3((unsigned)S), dim3(512), 0, stream, state_in, next, out, n, state_out,
                       seg_states, seg_words);
    return (int)hipGetLastError();
}

extern "C" int nerf_amd_launch_mt19937_uniform(const uint32_t* state_in, int next, float* out, long long n,
                                               uint32_t* state_out, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(mt19937_uniform_kernel<float>, dim3(1), dim3(512), 0, stream, state_in, next, out, n, state_out,
                       (const unsigned*)nullptr, 0ll);
    return (int)hipGetLastError();
}

// A draw that FOLLOWS a consumer nobody needs the numbers of (torch.rand(B, N) after torch.randperm(n): the reference's
// training iteration): the start states of its S segments -- segment 0 included, i.e. the generator's state after the
// skipped consumer -- come straight from the state BEFORE that consumer in ONE jump launch, jump distances summed:
// polys[b] = x^(624 (q + b * seg_blocks)) mod phi, seg_states[b] = the block 1 + q + b * seg_blocks after state_in's.
// (Jumping over the shuffle and then over the jitter's segments are two dependent launches of ~85 us each otherwise.)
// next1 = first unread word of seg_states[0]'s block.
extern "C" int nerf_amd_launch_mt19937_uniform_after(const uint32_t* state_in, const uint32_t* polys, int S, int next1,
                                                     float* out, long long n, uint32_t* state_out, long long seg_words,
                                                     uint32_t* seg_states, hipStream_t stream) {
    (void)hipGetLastError();
    const long long avail = MT_N - next1;
    long long need = 1;
    if (n > avail + seg_words) need = 1 + (n - avail - seg_words + seg_words - 1) / seg_words;
    if (need != S) return -1;
    hipLaunchKernelGGL(mt19937_zero_kernel, dim3((unsigned)S), dim3(MT_N), 0, stream, seg_states);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mt19937_jump_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       MT_JUMP_LDS_MAX);
    if (e != hipSuccess) return (int)e;
    const int lds = (MT_JUMP_BLOCKS + 2) * MT_N * (int)sizeof(unsigned);
    hipLaunchKernelGGL(mt19937_jump_kernel, dim3((unsigned)(S * MT_JUMP_SPLIT)), dim3(256), lds, stream, seg_states, polys, 0, 0, 0, 1, 1,
                       state_in);
    hipLaunchKernelGGL(mt19937_uniform_kernel<float>, dim3((unsigned)S), dim3(512), 0, stream, seg_states, next1, out, n, state_out,
                       seg_states, seg_words);
    return (int)hipGetLastError();
}

// The 32-bit outputs themselves (at::mt19937::operator(), CPUGeneratorImpl::random()): what torch.randperm draws.
extern "C" int nerf_amd_launch_mt19937_raw(const uint32_t* state_in, int next, uint32_t* out, long long n,
                                           uint32_t* state_out, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(mt19937_uniform_kernel<unsigned>, dim3(1), dim3(512), 0, stream, state_in, next, out, n, state_out,
                       (const unsigned*)nullptr, 0ll);
    return (int)hipGetLastError();
}

// state_out = the state words `1 + q` blocks after state_in's block, poly = x^(624 q) mod phi: what the generator holds
// after a consumer has drawn through that many regenerations (torch.randperm(n): n - 1 draws, of which only the first
// few are ever looked at -- csrc/select.hip).  state_out is an atomicXor target: zeroed here first.
extern "C" int nerf_amd_launch_mt19937_advance(const uint32_t* state_in, const uint32_t* poly, uint32_t* state_out,
                                               hipStream_t stream) {
    (void)hipGetLastError();
    // a kernel, not hipMemsetAsync: captured into a hipGraph a memset NODE in front of atomics has been seen to leave the
    // atomics on the buffer's old contents (csrc/dw_gemm.hip has the story); kernel -> kernel edges order correctly
    hipLaunchKernelGGL(mt19937_zero_kernel, dim3(1), dim3(MT_N), 0, stream, state_out);
    const int lds = (MT_JUMP_BLOCKS + 2) * MT_N * (int)sizeof(unsigned);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mt19937_jump_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            MT_JUMP_LDS_MAX);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mt19937_jump_kernel, dim3(MT_JUMP_SPLIT), dim3(256), lds, stream, state_out, poly, 0, 0, 0, 0, 1, state_in);
    return (int)hipGetLastError();
}
