#!/usr/bin/env python3
"""Model of the radix-4 stages of fft_big.hip at N = 16384 = 4^7 and N = 4096 = 4^6 (numpy, no GPU).
   tools/model_fft_big_r4.py [log2 N = 14 | 12] [rev]

The kernel's dataflow is the in-place binary one: layer l = 0 .. L-1 pairs the positions that differ in index bit L-1-l.  A
radix-4 DIF stage s (fft.h:311-349) is layers 2s and 2s + 1 with the reference's twiddle placement: after the first layer
the quarter (b1, b0) = (1, 1) is rotated by -+i (the temp2_timesi / temp4_timesi of fft.h:337-338), after the second layer the
quarters (0,1), (1,0), (1,1) are multiplied by W_G^(2n), W_G^(n), W_G^(3n) (the coefficients the reference applies when the next
stage loads its inputs, fft.h:322-335), G = N / 4^s, n = position mod G/4.  The result is X in BIT-reversed order (the
quarters (0,1) and (1,0) sit swapped against the reference's digit order, which only changes where the final permutation
puts them).  Every multiplier is formed the way the kernel forms it -- value of the per-thread table (capi.hip:
upload_thread_twiddles_big_r4, indexed exactly as there) times compile-time constant W_64^e -- and checked against the direct
formula.  Passes: 5 + 5 + R layers, R = L - 10; stage 2 straddles the first exchange."""
import sys

import numpy as np

L = int(sys.argv[1]) if len(sys.argv) > 1 else 14  # log2 N = 10 + R with R even: 14 or 12
N, T, R = 1 << L, (1 << L) // 32, L - 10
REV = len(sys.argv) > 2 and sys.argv[2] == "rev"
assert R in (2, 4)
sgn = 1.0 if REV else -1.0
row = np.exp(sgn * 2j * np.pi * np.arange(N) / N)  # the plan's row W_N^j, direction-folded


def W(G, e):
    return np.exp(sgn * 2j * np.pi * (np.asarray(e) % G) / G)


def rot(z):  # times -+i
    return z * (1j * sgn)


# the table of upload_thread_twiddles_big_r4: [slot < 14][thread t < T]
tab = np.zeros((14, T), complex)
for t in range(T):
    v, odd = t & ((1 << R) - 1), (t >> R) & 1
    for q in (1, 2, 3):
        tab[q - 1, t] = row[(q * t) % N]
        tab[2 + q, t] = row[(4 * q * t) % N]
        tab[7 + q, t] = row[(64 * q * v) % N]
        tab[10 + q, t] = row[(256 * q * v) % N]
    tab[6, t] = row[(16 * v) % N] if odd else row[0]
    tab[7, t] = row[(16 * (3 if odd else 2) * v) % N]

rng = np.random.default_rng(1)
x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
y = x.copy()
idx = np.arange(N)


def layer(y, b):
    lo = idx[(idx >> b) & 1 == 0]
    hi = lo | (1 << b)
    u, w = y[lo].copy(), y[hi].copy()
    y[lo], y[hi] = u + w, u - w


def bit(a, b):
    return (a >> b) & 1


def stage(y, reg, b1, thr_slots, const_unit, lowmask, G):
    """both layers on register bits b1, b1 - 1 of `reg` (index bits shift + b1, shift + b1 - 1), then the twiddles:
    table slot thr_slots[q - 1] of the element's thread x W_64^(q * (reg & lowmask) * const_unit); checked against W_G^(q n)"""
    shift = {id(k_of): L - 5, id(j_of): R, id(i_of): 0}[id(reg)]
    layer(y, shift + b1)
    m = (bit(reg, b1) == 1) & (bit(reg, b1 - 1) == 1)
    y[m] = rot(y[m])
    layer(y, shift + b1 - 1)
    for (q1, q0, q) in ((0, 1, 2), (1, 0, 1), (1, 1, 3)):
        m = (bit(reg, b1) == q1) & (bit(reg, b1 - 1) == q0)
        mult = W(64, q * (reg & lowmask) * const_unit)
        if thr_slots is not None:
            mult = mult * tab[thr_slots[q - 1], thread_of]
        assert np.allclose(mult[m], W(G, q * (idx % (G // 4)))[m]), (shift, b1, q)
        y[m] *= mult[m]


# ---------------- pass A: thread t = idx mod T holds k = idx / T (five bits)
t_of, k_of = idx % T, idx // T
j_of, i_of = (idx >> R) & 31, idx & 31
thread_of = t_of
stage(y, k_of, 4, (0, 1, 2), 2, 7, N)        # stage 0: W_N^(q t) x W_32^(q (k & 7))
stage(y, k_of, 2, (3, 4, 5), 8, 1, N // 4)   # stage 1: W_(N/4)^(q t) x W_8^(q (k & 1))
# stage 2, first layer: register bit 0 is index bit L-5, index bit L-6 is the thread's top bit
layer(y, L - 5)
m = ((k_of & 1) == 1) & (t_of >= T // 2)
y[m] = rot(y[m])

# ---------------- pass B: thread (blk = idx >> (5 + R), v = idx mod 2^R) holds j = (idx >> R) & 31
blk, v = idx >> (5 + R), idx & ((1 << R) - 1)
thread_of = (blk << R) | v
odd = blk & 1
# stage 2, second layer: register bit 4 (index bit L-6); quarter = (blk & 1, j >> 4); G = 2^(L-4), n = v + 2^R (j & 15)
layer(y, L - 6)
q_here = np.where(j_of < 16, np.where(odd == 1, 1, 0), np.where(odd == 1, 3, 2))
const = W(64, q_here * (j_of & 15))  # one of two literals, picked by the thread's parity
mult = np.where(j_of < 16, tab[6, thread_of], tab[7, thread_of]) * const
assert np.allclose(mult, W(1 << (L - 4), q_here * (idx % (1 << (L - 6)))))
y *= mult
stage(y, j_of, 3, (8, 9, 10), 4, 3, 1 << (R + 4))    # stage 3: W_(2^(R+4))^(q v) x W_16^(q (j & 3))
stage(y, j_of, 1, (11, 12, 13), 0, 0, 1 << (R + 2))  # stage 4: W_(2^(R+2))^(q v)

# ---------------- pass C: the last R layers on i = idx & 31 (constants only)
if R == 4:
    stage(y, i_of, 3, None, 4, 3, 16)  # W_16^(q (i & 3))
layer(y, 1)
m = (bit(i_of, 1) == 1) & (bit(i_of, 0) == 1)
y[m] = rot(y[m])
layer(y, 0)

brev = np.array([int(format(a, f"0{L}b")[::-1], 2) for a in range(N)])
X = np.fft.ifft(x) * N if REV else np.fft.fft(x)
err = np.abs(y - X[brev]).max() / np.abs(X).max()
print(f"N = {N}{' reverse' if REV else ''}: radix-4 layered dataflow against numpy.fft: max rel err {err:.2e}")
assert err < 1e-12
