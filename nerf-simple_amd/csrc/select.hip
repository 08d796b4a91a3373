// select.hip -- the first two lines of the reference's training loop on the device:
//     rays, ray_ids = rg.select(mode='train', N=batch_size)        train.py:47   (RayGenerator.select, utils/dataload.py:141-153:
//                                                                                ray_ids = torch.randperm(n)[:N]; rays = data[ray_ids, :])
//     gt_colors = train_imgs[ray_ids, :].float().cuda()           train.py:49
// with the ray table [n,6] and the colour table [n,3] resident in HBM (the reference keeps both in host memory, shuffles all
// n indices on one core -- 0.2 ... 0.7 s per iteration for 5 ... 16 M rays, against a 1.2 ms training step -- and copies
// the batch over PCIe).
//
// torch.randperm(n) on the CPU (ATen/native/TensorFactories.cpp randperm_cpu, n < 2^32 / 20) is a FORWARD Fisher-Yates
// shuffle of r = [0 .. n-1]:   for i in 0 .. n-2:  z = generator->random() % (n - i);  swap(r[i], r[i + z])
// -- one 32-bit MT19937 output per i, n - 1 in all.  Element i of the result is final after step i, so the first B
// elements need only the first B draws; the other n - 1 - B draws move the generator (csrc/host_rng.hip
// nerf_amd_mt19937_advance jumps over them) and nothing else.  And the first B steps touch at most 2 B positions, so
// they need no table of n entries either.  With j_i = i + z_i the swap partner of step i (j_i >= i):
//
//     a_i      = the value at position i before step i   = a_{pred(i)},  pred(i) = the LATEST k < i with j_k = i   (else i)
//     result_i = the value at position j_i before step i = a_{dup(i)},   dup(i)  = the LATEST k < i with j_k = j_i (else j_i)
//
// (a position p > i is only ever written by a step whose partner it is, with the value that step displaced from its own
// position; for j_i = i both lines say the same).  pred and dup are found by comparing against all earlier partners --
// B^2 / 2 comparisons, spread over B / 16 workgroups -- and the chains behind
// a_i are followed in the gather kernel (their expected length is B / n).  Exact for every (n, B), collisions included:
// oracle/nerf_oracle.py randperm_prefix is the sequential statement, pinned against torch.randperm itself.
//
// Two sources of z_i:  `draws` given -- the reference's numbers: the 32-bit outputs of torch's CPU generator, continued on
// the device (nerf_amd_mt19937_raw), so ray_ids and the generator afterwards are torch's, bit for bit;
// draws NULL -- the counter RNG (Philox-4x32-10 keyed by seed + *seed_mem): nothing to upload, and with seed_mem the step
// counter in device memory the launch sits inside a captured hipGraph and every replay selects a fresh batch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "nerf_device.h"

namespace {

constexpr int SEL_ROWS = 16;            // rows i per workgroup: a lane is (row = lane & 15, partner stream = lane >> 4)
constexpr int SEL_THREADS = 256;        // 4 waves x 4 lane groups = 16 partner streams over the same 16 rows
constexpr int SEL_CHUNK = 4096;         // partners staged in LDS at a time (16 KiB)
constexpr unsigned long long SEL_KEY = 0x73656c6563743a31ull;    // keeps the selection's Philox stream apart from the jitter's

__device__ __forceinline__ unsigned swap_partner(unsigned i, unsigned n, const unsigned* __restrict__ draws,
                                                 unsigned long long seed) {
    if (i + 1u >= n) return i;                                   // the last row of a full permutation draws nothing
    const unsigned z = draws ? draws[i] : philox_word(seed, i);
    return i + z % (n - i);                                      // n < 2^32 / 20: no overflow
}

// partner[i] = j_i for the B rows, once (one thread per row)
__global__ __launch_bounds__(256) void select_partner_kernel(const unsigned* __restrict__ draws, unsigned long long seed,
                                                             const unsigned long long* __restrict__ seed_mem, unsigned n, unsigned B,
                                                             unsigned* __restrict__ partner) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= B) return;
    seed ^= SEL_KEY;
    if (seed_mem) seed += *seed_mem;
    partner[i] = swap_partner(i, n, draws, seed);
}

// pred(i), dup(i) for 16 rows per workgroup against all earlier partners.  Work per workgroup: hi partners loaded from L2
// into LDS (16 bytes per thread and load) and hi x 16 comparisons pairs spread over 16 streams: every lane reads four
// consecutive partners of ITS stream (ds_read_b128: four addresses per wave-instruction, one per lane group), so one
// vector compare covers 16 rows x 4 partners.  (First form of this kernel: 64 rows per workgroup, every workgroup
// recomputing all earlier partners, one scalar broadcast per partner -- 32-38 us inside a training step, 3 % of it: a
// wave64 vector op is 4 cycles, and 1024 partners x 6 ops per wave is 25 k cycles.  This form: profiles/r04_select_kernels.txt.)
__global__ __launch_bounds__(SEL_THREADS) void select_scan_kernel(unsigned B, const unsigned* __restrict__ partner,
                                                                  int* __restrict__ pred, int* __restrict__ dup) {
    __shared__ __attribute__((aligned(16))) unsigned js[SEL_CHUNK];
    __shared__ int pred_s[SEL_ROWS], dup_s[SEL_ROWS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row = lane & 15, stream = wave * 4 + (lane >> 4);
    const unsigned base = blockIdx.x * SEL_ROWS;
    const unsigned i = base + row;
    const unsigned hi = min(base + SEL_ROWS, B);                 // partners of rows < hi are all this workgroup looks at
    const unsigned my_j = i < B ? partner[i] : 0xffffffffu;
    if (t < SEL_ROWS) {
        pred_s[t] = -1;
        dup_s[t] = -1;
    }
    int p = -1, d = -1;
    for (unsigned c0 = 0; c0 < hi; c0 += SEL_CHUNK) {
        const unsigned cn = min((unsigned)SEL_CHUNK, hi - c0);
        __syncthreads();                                         // the previous chunk has been read
        for (unsigned k = 4u * t; k < cn; k += 4u * SEL_THREADS) {           // the workspace is 16-byte aligned and padded
            const uint4 v = *reinterpret_cast<const uint4*>(partner + c0 + k);
            *reinterpret_cast<uint4*>(js + k) = v;
        }
        __syncthreads();
        // stream s takes the 16-byte blocks s, s + 16, s + 32, ...: ascending inside a lane, so its last match is its latest
        for (unsigned k0 = 4u * stream; k0 < cn; k0 += 64u) {
            const uint4 v = *reinterpret_cast<const uint4*>(js + k0);
            const unsigned jk[4] = {v.x, v.y, v.z, v.w};
            const unsigned kk = c0 + k0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool live = kk + e < i && k0 + e < cn;     // an earlier row's partner (padding beyond cn is never live)
                if (live && jk[e] == i) p = (int)(kk + e);
                if (live && jk[e] == my_j) d = (int)(kk + e);
            }
        }
    }
    // latest match over the 16 streams: the 4 lane groups of a wave by lane exchange, the 4 waves through LDS
    p = max(p, __shfl_xor(p, 16));
    p = max(p, __shfl_xor(p, 32));
    d = max(d, __shfl_xor(d, 16));
    d = max(d, __shfl_xor(d, 32));
    if (lane < 16) {
        if (p >= 0) atomicMax(&pred_s[row], p);
        if (d >= 0) atomicMax(&dup_s[row], d);
    }
    __syncthreads();
    if (t < SEL_ROWS && i < B) {
        pred[i] = pred_s[t];
        dup[i] = dup_s[t];
    }
}

__global__ __launch_bounds__(256) void select_gather_kernel(const unsigned* __restrict__ partner, const int* __restrict__ pred,
                                                            const int* __restrict__ dup, unsigned B,
                                                            const float* __restrict__ table, const float* __restrict__ colours,
                                                            float* __restrict__ rays_out, float* __restrict__ gt_out,
                                                            long long* __restrict__ ids_out) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= B) return;
    long long id;
    int k = dup[i];
    if (k < 0) {
        id = partner[i];
    } else {
        for (int up = pred[k]; up >= 0; up = pred[k]) k = up;    // a_k: follow the writers of position k back to an untouched one
        id = k;
    }
    if (ids_out) ids_out[i] = id;
    if (rays_out && table) {
        const float2* src = reinterpret_cast<const float2*>(table + id * 6);        // rows of 24 B: 8-byte aligned
        float2* dst = reinterpret_cast<float2*>(rays_out + (long long)i * 6);
        const float2 a = src[0], b = src[1], c = src[2];
        dst[0] = a;
        dst[1] = b;
        dst[2] = c;
    }
    if (gt_out && colours) {
        const float* src = colours + id * 3;
        float* dst = gt_out + (long long)i * 3;
        const float r = src[0], g = src[1], b = src[2];
        dst[0] = r;
        dst[1] = g;
        dst[2] = b;
    }
}

// ---- x^J mod phi on the host (GF(2)[x], 64-bit limbs) -------------------------------------------------------------
constexpr int DEG = 19937;
constexpr int LIMBS = 312;                                       // 19968 bits hold a residue (degree < 19937)

struct Poly {
    uint64_t w[2 * LIMBS + 1];
    void clear() { memset(w, 0, sizeof(w)); }
    bool bit(int i) const { return (w[i >> 6] >> (i & 63)) & 1u; }
};

inline uint64_t spread32(uint32_t x) {                           // bit i -> bit 2i
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

// r ^= phi << s for every set bit DEG + s of r, from the top down: r mod phi
void reduce(Poly& r, const uint64_t (*phi_sh)[LIMBS + 1], int top_bit) {
    for (int d = top_bit; d >= DEG; --d) {
        if (!r.bit(d)) continue;
        const int s = d - DEG, limb = s >> 6;
        const uint64_t* ph = phi_sh[s & 63];
        for (int k = 0; k <= LIMBS; ++k) r.w[limb + k] ^= ph[k];
    }
}

}  // namespace

extern "C" int nerf_amd_launch_select_rays(const uint32_t* draws, unsigned long long seed, const unsigned long long* seed_mem,
                                           long long n, long long B, const float* table, const float* colours, float* rays_out,
                                           float* gt_out, long long* ids_out, void* workspace, hipStream_t stream) {
    (void)hipGetLastError();
    // workspace: partner[B rounded up to 4] | pred[B] | dup[B]  (nerf_amd_select_workspace_bytes)
    const long long Bp = (B + 3) / 4 * 4;
    unsigned* partner = reinterpret_cast<unsigned*>(workspace);
    int* pred = reinterpret_cast<int*>(partner + Bp);
    int* dup = pred + B;
    hipLaunchKernelGGL(select_partner_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream, draws, seed, seed_mem,
                       (unsigned)n, (unsigned)B, partner);
    const unsigned rows = (unsigned)((B + SEL_ROWS - 1) / SEL_ROWS);
    hipLaunchKernelGGL(select_scan_kernel, dim3(rows), dim3(SEL_THREADS), 0, stream, (unsigned)B, partner, pred, dup);
    hipLaunchKernelGGL(select_gather_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream, partner, pred, dup,
                       (unsigned)B, table, colours, rays_out, gt_out, ids_out);
    return (int)hipGetLastError();
}

// h_out624 = x^(624 * blocks) mod phi as 624 little-endian 32-bit words; h_phi624 = phi (degree 19937, utils/mt19937_jump.npz).
// Square-and-multiply from the top bit of the exponent; a multiplication by x is a shift.  ~0.1 s for a 30-bit exponent.
extern "C" int nerf_amd_host_mt19937_jump_poly(long long blocks, const uint32_t* h_phi624, uint32_t* h_out624) {
    const unsigned long long J = (unsigned long long)blocks * 624ull;
    uint64_t phi[LIMBS + 1] = {};
    for (int i = 0; i < 624; ++i) phi[i >> 1] |= (uint64_t)h_phi624[i] << (32 * (i & 1));
    if (!((phi[DEG >> 6] >> (DEG & 63)) & 1u) || (phi[DEG >> 6] >> ((DEG & 63) + 1)) != 0 || !(phi[0] & 1u)) return -1;   // not phi
    std::vector<uint64_t> sh(64 * (LIMBS + 1));
    uint64_t (*phi_sh)[LIMBS + 1] = reinterpret_cast<uint64_t (*)[LIMBS + 1]>(sh.data());
    for (int s = 0; s < 64; ++s) {
        for (int k = 0; k <= LIMBS; ++k) phi_sh[s][k] = 0;
        for (int k = 0; k < LIMBS; ++k) {
            phi_sh[s][k] |= phi[k] << s;
            if (s) phi_sh[s][k + 1] |= phi[k] >> (64 - s);
        }
    }
    std::vector<Poly> store(2);
    Poly &r = store[0], &sq = store[1];
    r.clear();
    r.w[0] = 1;
    int nbits = 0;
    while (nbits < 64 && (J >> nbits)) ++nbits;
    for (int b = nbits - 1; b >= 0; --b) {
        sq.clear();
        for (int k = 0; k < LIMBS; ++k) {
            sq.w[2 * k] = spread32((uint32_t)r.w[k]);
            sq.w[2 * k + 1] = spread32((uint32_t)(r.w[k] >> 32));
        }
        reduce(sq, phi_sh, 2 * (DEG - 1));
        if ((J >> b) & 1ull) {
            uint64_t carry = 0;
            for (int k = 0; k <= LIMBS; ++k) {
                const uint64_t v = sq.w[k];
                sq.w[k] = (v << 1) | carry;
                carry = v >> 63;
            }
            reduce(sq, phi_sh, DEG);
        }
        memcpy(r.w, sq.w, sizeof(r.w));
    }
    for (int i = 0; i < 624; ++i) h_out624[i] = (uint32_t)(r.w[i >> 1] >> (32 * (i & 1)));
    return 0;
}
