"""Jump-ahead polynomials of MT19937 for utils/host_rng.py (writes nerf-simple_amd/utils/mt19937_jump.npz).

MT19937's one-word step F is linear over GF(2) on a 19937-bit state, so F^J = g_J(F) with
g_J(x) = x^J mod phi(x), phi the characteristic polynomial of F (Haramoto, Matsumoto, Nishimura,
Panneton, L'Ecuyer: "Efficient jump ahead for F2-linear random number generators", 2008).  With
w_0, w_1, ... the raw (untempered) word sequence that starts with the state's own 624 words,
F^i applied to the state is the window w_i .. w_{i+623}, hence

        (F^J s)[k] = XOR over the set bits i of g_J of w[i + k],        k = 0 .. 623

-- a GF(2) convolution over 19937 + 624 words that csrc/host_rng.hip evaluates in parallel.
phi is found with Berlekamp-Massey on one output bit (Python integers as bit vectors), g_J by
square-and-multiply.  Polynomials for J = SEG_WORDS * 2^m, m = 0 .. M-1, are stored so that the
start states of up to 2^M segments follow from the first by a doubling tree; for draws shorter than one
few such segments a finer cut is stored with the polynomial of every multiple (J = SHORT_SEG_WORDS * j, j = 1 .. 63).

Everything here is checked again by tests/test_oracle_golden.py (phi recomputed, a jump compared
with plain sequential generation)."""
import os

import numpy as np

N, M_ = 624, 397
DEG = 19937
SEG_BLOCKS = 1024                      # a segment = 1024 blocks of 624 words = 638,976 draws
SEG_WORDS = SEG_BLOCKS * N
LEVELS = 10                            # up to 1024 segments (6.5e8 draws) per call
# the draws of a training batch (4096 rays x 64 .. 128 samples = 2.6e5 .. 5.2e5) fit ONE such segment, those of the
# reference's test batch (16,000 rays x 128 = 2.0e6) four: a second, finer cut with the polynomial of every multiple
# stored, so that all start states follow from the first in ONE launch
SHORT_SEG_BLOCKS = 64                  # 39,936 draws
SHORT_SEG_WORDS = SHORT_SEG_BLOCKS * N
SHORT_COUNT = 63                       # x^(j * SHORT_SEG_WORDS), j = 1 .. 63: up to 64 segments = 2.56e6 draws


def raw_words(state, nblocks):
    """The raw word sequence w_0.. : the state's 624 words followed by nblocks regenerated blocks."""
    mt = np.asarray(state, dtype=np.uint32).copy()
    out = [mt.copy()]
    D = N - M_

    def twist(u, v):
        y = (u & np.uint32(0x80000000)) | (v & np.uint32(0x7fffffff))
        return (y >> np.uint32(1)) ^ np.where(v & np.uint32(1), np.uint32(0x9908b0df), np.uint32(0)).astype(np.uint32)

    for _ in range(nblocks):
        new = mt.copy()
        new[:D] = mt[M_:] ^ twist(mt[:D], mt[1:D + 1])
        for lo in range(D, N - 1, D):
            hi = min(lo + D, N - 1)
            new[lo:hi] = new[lo - D:hi - D] ^ twist(mt[lo:hi], mt[lo + 1:hi + 1])
        new[N - 1] = new[M_ - 1] ^ twist(mt[N - 1:N], new[0:1])[0]
        mt = new
        out.append(mt.copy())
    return np.concatenate(out)


def berlekamp_massey(bits):
    """Connection polynomial C (int, bit i = c_i, c_0 = 1) and linear complexity L of a GF(2) sequence:
    s_n = XOR_{i=1..L} c_i s_{n-i}."""
    C, B, L, m = 1, 1, 0, 1
    rev = 0                                         # bit i = s_{n-i}
    for n, s in enumerate(bits):
        rev = (rev << 1) | int(s)
        d = (C & rev).bit_count() & 1
        if d == 0:
            m += 1
        elif 2 * L <= n:
            T = C
            C ^= B << m
            L, B, m = n + 1 - L, T, 1
        else:
            C ^= B << m
            m += 1
    return C, L


def char_poly():
    """phi(x) of the one-word step, as an int (bit i = coefficient of x^i), degree 19937."""
    rng = np.random.default_rng(12345)
    state = rng.integers(0, 2 ** 32, size=N, dtype=np.uint64).astype(np.uint32)
    w = raw_words(state, 2 * DEG // N + 3)
    bits = (w[:2 * DEG + 64] & np.uint32(1)).tolist()
    C, L = berlekamp_massey(bits)
    assert L == DEG, L
    # s_n = sum c_i s_{n-i}  <=>  the sequence is annihilated by the reciprocal polynomial x^L C(1/x)
    phi = 0
    for i in range(L + 1):
        if (C >> i) & 1:
            phi |= 1 << (L - i)
    return phi


def poly_mod(a, phi):
    dp = phi.bit_length() - 1
    while a.bit_length() - 1 >= dp:
        a ^= phi << (a.bit_length() - 1 - dp)
    return a


def poly_square(a):
    return int("0".join(bin(a)[2:]), 2)             # spread the bits: (sum a_i x^i)^2 = sum a_i x^(2i)


def x_pow_mod(J, phi):
    """x^J mod phi by square-and-multiply (multiplying by x is a shift)."""
    r = 1
    for bit in bin(J)[2:]:
        r = poly_mod(poly_square(r), phi)
        if bit == "1":
            r = poly_mod(r << 1, phi)
    return r


def to_words(p):
    return np.array([(p >> (32 * i)) & 0xffffffff for i in range(N)], dtype=np.uint32)


def apply_jump(state, gwords):
    """(F^J s) for a block-aligned state s (numpy restatement of the convolution)."""
    w = raw_words(state, DEG // N + 2)              # w_0 .. w_{>= 19937 + 623}
    out = np.zeros(N, dtype=np.uint32)
    g = int.from_bytes(np.asarray(gwords, dtype="<u4").tobytes(), "little")
    i = 0
    while g:
        if g & 1:
            out ^= w[i:i + N]
        g >>= 1
        i += 1
    return out


if __name__ == "__main__":
    phi = char_poly()
    polys = np.stack([to_words(x_pow_mod(SEG_WORDS << m, phi)) for m in range(LEVELS)])
    short_polys = np.stack([to_words(x_pow_mod(SHORT_SEG_WORDS * j, phi)) for j in range(1, SHORT_COUNT + 1)])
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf-simple_amd", "utils",
                       "mt19937_jump.npz")
    np.savez_compressed(dst, seg_words=np.int64(SEG_WORDS), polys=polys, phi=to_words(phi),
                        phi_top=np.int64(phi >> (32 * N)), short_seg_words=np.int64(SHORT_SEG_WORDS), short_polys=short_polys)
    print("wrote", dst, polys.shape, short_polys.shape, "phi weight", phi.bit_count())
