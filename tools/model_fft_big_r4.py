#!/usr/bin/env python3
"""Model of the radix-4 stages of fft_big.hip at N = 16384 = 4^7 (numpy, no GPU).

The kernel's dataflow is the in-place binary one: layer l = 0..13 pairs the positions that differ in index bit 13 - l.  A
radix-4 DIF stage s (fft.h:311-349) is layers 2s and 2s + 1 with the reference's twiddle placement: after the first layer
the quarter (b1, b0) = (1, 1) is rotated by -+i (the temp2_timesi / temp4_timesi of fft.h:337-338), after the second layer the
quarters (0,1), (1,0), (1,1) are multiplied by W_G^(2n), W_G^(n), W_G^(3n) (the coefficients the reference applies when the next
stage loads its inputs, fft.h:322-335), G = N / 4^s, n = position mod G/4.  The result is X in BIT-reversed order (the
quarters (0,1) and (1,0) sit swapped against the reference's digit order, which only changes where the final permutation
puts them).  Every multiplier is formed the way the kernel forms it -- per-thread table value times compile-time constant --
and checked against the direct formula; the passes are 5 + 5 + 4 layers, stage 2 straddles the first exchange."""
import numpy as np

L, N = 14, 1 << 14
T = 512
REV = False
sgn = 1.0 if REV else -1.0


def W(G, e):
    return np.exp(sgn * 2j * np.pi * (e % G) / G)


def rot(z):  # times -+i
    return z * (1j * sgn)


rng = np.random.default_rng(1)
x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
y = x.copy()
idx = np.arange(N)


def layer(y, b):
    lo = idx[(idx >> b) & 1 == 0]
    hi = lo | (1 << b)
    u, w = y[lo].copy(), y[hi].copy()
    y[lo], y[hi] = u + w, u - w


def bits(a, hi, lo):
    return (a >> lo) & ((1 << (hi - lo + 1)) - 1)


# ---------------- pass A: thread t < 512 holds k = idx >> 9 (five bits), idx = t + 512 k
t_of, k_of = idx & 511, idx >> 9
# thread twiddles of pass A: [stage 0: W_N^(q t)], [stage 1: W_4096^(q t)], q = 1..3
thrA0 = {q: W(N, q * t_of) for q in (1, 2, 3)}
thrA1 = {q: W(4096, q * t_of) for q in (1, 2, 3)}
# stage 0: layers 0, 1 = k bits 4, 3
layer(y, 13)
m = (bits(k_of, 4, 4) == 1) & (bits(k_of, 3, 3) == 1)
y[m] = rot(y[m])
layer(y, 12)
for (b1, b0, q) in ((0, 1, 2), (1, 0, 1), (1, 1, 3)):
    m = (bits(k_of, 4, 4) == b1) & (bits(k_of, 3, 3) == b0)
    kk = k_of & 7
    mult = thrA0[q] * W(32, q * kk)  # thread value x constant W_32^(q kk)
    assert np.allclose(mult[m], W(N, q * (idx % 4096))[m])
    y[m] *= mult[m]
# stage 1: layers 2, 3 = k bits 2, 1
layer(y, 11)
m = (bits(k_of, 2, 2) == 1) & (bits(k_of, 1, 1) == 1)
y[m] = rot(y[m])
layer(y, 10)
for (b1, b0, q) in ((0, 1, 2), (1, 0, 1), (1, 1, 3)):
    m = (bits(k_of, 2, 2) == b1) & (bits(k_of, 1, 1) == b0)
    mult = thrA1[q] * W(8, q * (k_of & 1))
    assert np.allclose(mult[m], W(4096, q * (idx % 1024))[m])
    y[m] *= mult[m]
# stage 2, first layer: k bit 0 (index bit 9); its -+i goes to (bit 9, bit 8) = (1, 1): k odd and t >= 256
layer(y, 9)
m = ((k_of & 1) == 1) & (t_of >= 256)
y[m] = rot(y[m])

# ---------------- pass B: thread (blk = idx >> 9, v = idx & 15) holds j = (idx >> 4) & 31
blk, v, j = idx >> 9, idx & 15, (idx >> 4) & 31
odd = blk & 1
# stage 2, second layer: j bit 4 (index bit 8); quarter = (blk & 1, j >> 4)
layer(y, 8)
n2 = v + 16 * (j & 15)
qa = np.where(odd == 1, 1, 0)  # j < 16
qb = np.where(odd == 1, 3, 2)  # j >= 16
thr2a, thr2b = W(1024, qa * v), W(1024, qb * v)  # two table values per thread
q_here = np.where(j < 16, qa, qb)
const = W(64, q_here * (j & 15))  # the constant is one of two literals, picked by the thread's parity
mult = np.where(j < 16, thr2a, thr2b) * const
assert np.allclose(mult, W(1024, q_here * n2))
y *= mult
# stage 3: layers 6, 7 = j bits 3, 2; G = 256, n = v + 16 (j & 3)
thrB3 = {q: W(256, q * v) for q in (1, 2, 3)}
layer(y, 7)
m = (bits(j, 3, 3) == 1) & (bits(j, 2, 2) == 1)
y[m] = rot(y[m])
layer(y, 6)
for (b1, b0, q) in ((0, 1, 2), (1, 0, 1), (1, 1, 3)):
    m = (bits(j, 3, 3) == b1) & (bits(j, 2, 2) == b0)
    mult = thrB3[q] * W(16, q * (j & 3))
    assert np.allclose(mult[m], W(256, q * (idx % 64))[m])
    y[m] *= mult[m]
# stage 4: layers 8, 9 = j bits 1, 0; G = 64, n = v
thrB4 = {q: W(64, q * v) for q in (1, 2, 3)}
layer(y, 5)
m = (bits(j, 1, 1) == 1) & (bits(j, 0, 0) == 1)
y[m] = rot(y[m])
layer(y, 4)
for (b1, b0, q) in ((0, 1, 2), (1, 0, 1), (1, 1, 3)):
    m = (bits(j, 1, 1) == b1) & (bits(j, 0, 0) == b0)
    y[m] *= thrB4[q][m]

# ---------------- pass C: thread w holds i = idx & 31; stages 5, 6 on i bits 3..0 (constants only)
i = idx & 31
layer(y, 3)
m = (bits(i, 3, 3) == 1) & (bits(i, 2, 2) == 1)
y[m] = rot(y[m])
layer(y, 2)
for (b1, b0, q) in ((0, 1, 2), (1, 0, 1), (1, 1, 3)):
    m = (bits(i, 3, 3) == b1) & (bits(i, 2, 2) == b0)
    y[m] *= W(16, q * (i & 3))[m]
layer(y, 1)
m = (bits(i, 1, 1) == 1) & (bits(i, 0, 0) == 1)
y[m] = rot(y[m])
layer(y, 0)

# y[idx] = X[bit_reverse14(idx)]
brev = np.array([int(format(a, "014b")[::-1], 2) for a in range(N)])
X = np.fft.ifft(x) * N if REV else np.fft.fft(x)
err = np.abs(y - X[brev]).max() / np.abs(X).max()
print(f"N = {N}: radix-4 layered dataflow against numpy.fft: max rel err {err:.2e}")
assert err < 1e-12
