#!/usr/bin/env python3
"""tools/model_fft_wave.py -- LDS slot map of csrc/fft_wave.hip (N = 1024, one transform per wave).

A wave exchanges its 1024 points through LDS twice; the three access patterns (8-byte units, lane t, register index):
  A  p = t + 64 k                  written after pass A
  B  p = 64 (t >> 2) + (t & 3) + 4 j   read before / written after pass B
  C  p = 16 w(t) + i               read before pass C, w = bit_reverse6(t) (radix 2) or digit_reverse4(t) (radix 4)
ds_read/write_b64 are serviced per half-wave over 32 bank pairs (slot mod 32).  Slots are p ^ X(p >> 5) with X linear over
GF(2) (five rows of five bits): this script searches rows for which every pattern of BOTH radices puts the 32 lanes of each
half-wave on 32 distinct bank pairs, and verifies the rows compiled into the kernel (kRow).
"""
import random

K_ROW = [2, 30, 15, 25, 26]  # csrc/fft_wave.hip: rows<5> (float2: 8-byte units, X of p >> 5)
K_ROW64 = [7, 15, 8, 15, 11, 14]  # rows<4> (double2: 16-byte units, X of p >> 4; ds_*_b128: quarter-waves over 16 bank quads)


def brev6(t):
    return int(f"{t:06b}"[::-1], 2)


def drev3(t):
    return ((t & 3) << 4) | (t & 12) | ((t >> 4) & 3)


def patterns(radix):
    pats = [[t + 64 * k for t in range(64)] for k in range(16)]
    pats += [[64 * (t >> 2) + (t & 3) + 4 * j for t in range(64)] for j in range(16)]
    w = brev6 if radix == 2 else drev3
    pats += [[16 * w(t) + i for t in range(64)] for i in range(16)]
    return pats


def slot(rows, p, sh=5):
    x = 0
    for b in range(10 - sh):
        if (p >> (sh + b)) & 1:
            x ^= rows[b]
    return p ^ x


def conflict_free(rows, pats, sh=5):
    lanes = 1 << sh  # 32 lanes over 32 bank pairs (b64) / 16 lanes over 16 bank quads (b128)
    return all(len({slot(rows, p, sh) & (lanes - 1) for p in pat[h:h + lanes]}) == lanes for pat in pats for h in range(0, 64, lanes))


# ---- N = 256 / 512 / 2048 (sdsp_fft_wave_f32): P = N / 64 points per lane, passes of log2 P radix-2 stages
ROWS2 = {8: [31, 20, 10], 9: [23, 31, 20, 26], 11: [9, 15, 24, 2, 5, 6]}  # csrc/fft_wave.hip: rows2<L>


def patterns2(L):
    LP = L - 6
    P, N, NP = 1 << LP, 1 << L, -(-L // LP)
    pats = []
    for i in range(NP - 1):  # pass i: positions b (s P) + v + s k
        sg = N >> (LP * (i + 1))
        pats += [[(t // sg) * (sg * P) + (t % sg) + sg * k for t in range(64)] for k in range(P)]
    pats += [[P * brev6(t) + k for t in range(64)] for k in range(P)]  # last pass: P contiguous positions of block w
    return pats


def slot2(rows, p, L):
    x = 0
    for b in range(L - 5):
        if (p >> (5 + b)) & 1:
            x ^= rows[b]
    return p ^ x


def conflict_free2(rows, pats, L):
    return all(len({slot2(rows, p, L) & 31 for p in pat[h:h + 32]}) == 32 for pat in pats for h in (0, 32))


if __name__ == "__main__":
    for L, rows in ROWS2.items():
        pats = patterns2(L)
        assert all(sorted(p for pat in pats[i * (1 << (L - 6)):(i + 1) * (1 << (L - 6))] for p in pat) == list(range(1 << L))
                   for i in range(len(pats) >> (L - 6))), "a pass layout does not cover the transform"
        assert conflict_free2(rows, pats, L), f"rows2<{L}> are not conflict free"
        assert sorted(slot2(rows, p, L) for p in range(1 << L)) == list(range(1 << L))
        print(f"rows2<{L}> = {rows}: {len(pats)} access patterns conflict free")
    p2, p4 = patterns(2), patterns(4)
    assert conflict_free(K_ROW, p2) and conflict_free(K_ROW, p4), "the compiled rows are not conflict free"
    ident = [0, 0, 0, 0, 0]
    worst = max(32 // len({p & 31 for p in pat[:32]}) for pat in p2 + p4)
    print(f"kRow = {K_ROW}: conflict free for both radices (no swizzle: up to {worst}-way conflicts)")
    # every slot is used exactly once
    assert sorted(slot(K_ROW, p) for p in range(1024)) == list(range(1024))
    assert conflict_free(K_ROW64, p2, 4) and conflict_free(K_ROW64, p4, 4), "the compiled f64 rows are not conflict free"
    assert sorted(slot(K_ROW64, p, 4) for p in range(1024)) == list(range(1024))
    print(f"rows<4> = {K_ROW64}: conflict free for both radices (f64, quarter-waves)")
    random.seed(1)
    for it in range(100000):
        rows = [random.randrange(32) for _ in range(5)]
        if conflict_free(rows, p2) and conflict_free(rows, p4):
            print(f"search: rows {rows} after {it + 1} draws")
            break
