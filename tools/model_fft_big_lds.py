#!/usr/bin/env python3
"""The LDS byte addresses fft_big.hip uses for its three access patterns against the generic swizzle sw<L>() they replace,
for every thread and register of N = 8192 / 16384 / 32768 (pure Python, no GPU)."""


def sw(L, p):
    R = L - 10
    return p ^ ((((p >> (5 + R)) & ((1 << (5 - R)) - 1)) << R) | ((p >> 10) & ((1 << R) - 1)))


def rot5(R, k):
    return ((k & ((1 << (5 - R)) - 1)) << R) | ((k >> (5 - R)) & ((1 << R) - 1))


def brev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2)


for L in (11, 12, 13, 14, 15):
    R = L - 10
    N = 1 << L
    T = M = N // 32
    JL = 1 << (5 - R)
    for t in range(T):
        blk, v = t >> R, t & ((1 << R) - 1)
        pb = blk * M + v
        xb = rot5(R, blk)
        base_b = [4 * (blk * M + (v ^ (xb & ((1 << R) - 1))) + ((jl ^ (xb >> R)) << R)) for jl in range(JL)]
        w = brev(t, L - 5)
        base_c = (128 * w) | (4 * ((((w >> R) & ((1 << (5 - R)) - 1)) << R) | ((w >> 5) & ((1 << R) - 1))))
        for k in range(32):
            assert 4 * sw(L, k * M + t) == 4 * k * M + ((4 * t) ^ (4 * rot5(R, k)))        # pattern A
            assert 4 * sw(L, pb + (k << R)) == base_b[k % JL] + 128 * (k // JL)             # pattern B
            assert 4 * sw(L, 32 * w + k) == base_c ^ (4 * k)                                # pattern C
    print(f"N = {N}: patterns A, B, C equal sw<{L}> for all {T} threads x 32 registers")
