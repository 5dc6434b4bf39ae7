#!/usr/bin/env python3
"""tools/model_fft_wave.py -- LDS slot maps of csrc/fft_wave.hip (one wave per 1024 points / per transform).

A wave exchanges its points through LDS between register passes.  Slots are p ^ X(p >> SH) with X linear over GF(2); this
script verifies the rows compiled into the kernels against the banking of MI355X_MICROARCH.md (section LDS) and can search
new ones:
  ds_read_b64    2 x 32 lanes {0-31}, {32-63}; bank (a / 4) mod 64   -> 8-byte slots distinct mod 32 per group
  ds_write_b64   4 x 16 contiguous lanes;      bank (a / 4) mod 32   -> 8-byte slots distinct mod 16 per group
  ds_read_b128   4 x 16 lanes in the guide's groups; mod 64          -> 16-byte slots distinct mod 16 per group
  ds_write_b128  8 x 8 contiguous lanes;       mod 32                -> 16-byte slots distinct mod 8 per group
(The first maps of round 2 assumed the read rule for the writes as well: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE read 25 %
at N = 1024 and 30 % at N = 256; profiles/r02_lds_bank_conflicts.md.)
"""
import itertools
import random
import sys

G_R64 = [list(range(0, 32)), list(range(32, 64))]
G_W64 = [list(range(h, h + 16)) for h in range(0, 64, 16)]
_a = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
_b = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
G_R128 = [_a, _b, [l + 32 for l in _a], [l + 32 for l in _b]]
G_W128 = [list(range(h, h + 8)) for h in range(0, 64, 8)]


def brev(v, n):
    return int(f"{v:0{n}b}"[::-1], 2) if n else 0


def drev3(t):
    return ((t & 3) << 4) | (t & 12) | ((t >> 4) & 3)


def layouts_1024(radix):
    """N = 1024 (sdsp_fft1024_wave): layouts A, B, C; writes in A and B, reads in B and C."""
    A = [[t + 64 * k for t in range(64)] for k in range(16)]
    B = [[64 * (t >> 2) + (t & 3) + 4 * j for t in range(64)] for j in range(16)]
    w = (lambda t: brev(t, 6)) if radix == 2 else drev3
    C = [[16 * w(t) + i for t in range(64)] for i in range(16)]
    return B + C, A + B


def layouts_p(L, radix=2):
    """N = 256 / 512 / 2048 (sdsp_fft_wave_f32): pass i < last: b (s P) + v + s k; last: P w + k, w = bit_reverse6(t) or (radix 4,
    N = 256) digit_reverse4(t); reads / writes."""
    LP = L - 6
    P, N, NP = 1 << LP, 1 << L, -(-L // LP)
    lay = []
    for i in range(NP - 1):
        sg = N >> (LP * (i + 1))
        lay.append([[(t // sg) * (sg * P) + (t % sg) + sg * k for t in range(64)] for k in range(P)])
    lay.append([[P * (brev(t, 6) if radix == 2 else drev3(t)) + k for t in range(64)] for k in range(P)])
    return [p for l in lay[1:] for p in l], [p for l in lay[:-1] for p in l]


def slot(rows, p, sh):
    x = 0
    for b, r in enumerate(rows):
        if (p >> (sh + b)) & 1:
            x ^= r
    return p ^ x


def free(rows, sh, reads, writes, g_r, m_r, g_w, m_w):
    return (all(len({slot(rows, pat[l], sh) % m_r for l in g}) == len(g) for pat in reads for g in g_r) and
            all(len({slot(rows, pat[l], sh) % m_w for l in g}) == len(g) for pat in writes for g in g_w))


CASES = {  # name: (rows, SH, N, [(reads, writes), ...], read groups / modulus, write groups / modulus)
    "rows<5>  N = 1024 f32, both radices": ([16, 29, 6, 23, 18], 5, 1024, [layouts_1024(2), layouts_1024(4)], G_R64, 32, G_W64, 16),
    "rows<4>  N = 1024 f64, both radices": ([5, 3, 15, 1, 6, 9], 4, 1024, [layouts_1024(2), layouts_1024(4)], G_R128, 16, G_W128, 8),
    "rows2<8>  N = 256, both radices": ([4, 9, 17, 2], 4, 256, [layouts_p(8), layouts_p(8, 4)], G_R64, 32, G_W64, 16),
    "rows2<9>  N = 512": ([10, 28, 15, 18], 5, 512, [layouts_p(9)], G_R64, 32, G_W64, 16),
    "rows2<11> N = 2048": ([25, 23, 13, 5, 7, 30], 5, 2048, [layouts_p(11)], G_R64, 32, G_W64, 16),
}

if __name__ == "__main__":
    for name, (rows, sh, n, sets, g_r, m_r, g_w, m_w) in CASES.items():
        assert sorted(slot(rows, p, sh) for p in range(n)) == list(range(n)), name
        assert all(free(rows, sh, r, w, g_r, m_r, g_w, m_w) for r, w in sets), f"{name}: not conflict free"
        ident = [0] * len(rows)
        worst = 1
        for r, w in sets:
            for pats, groups, m in ((r, g_r, m_r), (w, g_w, m_w)):
                for pat in pats:
                    for g in groups:
                        banks = [pat[l] % m for l in g]
                        worst = max(worst, max(banks.count(b) for b in set(banks)))
        print(f"{name}: {rows} conflict free (unswizzled: up to {worst}-way)")
    if len(sys.argv) > 1 and sys.argv[1] == "search":  # how the rows were found
        random.seed(5)
        for name, (rows, sh, n, sets, g_r, m_r, g_w, m_w) in CASES.items():
            top = 1 << sh
            for it in range(2000000):
                cand = [random.randrange(top if b or sh == 5 else 16) for b in range(len(rows))]
                if all(free(cand, sh, r, w, g_r, m_r, g_w, m_w) for r, w in sets):
                    print(f"search {name}: {cand} after {it + 1} draws")
                    break
