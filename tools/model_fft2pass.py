#!/usr/bin/env python3
"""CPU model of csrc/fft_2pass.hip: the kernels' index arithmetic (thread <-> element maps, LDS slot maps with their XOR
swizzles, intermediate layout, output scatter) replayed in numpy, in double, against numpy.fft.  Catches mapping mistakes
before a GPU sees the kernel; says nothing about bank conflicts or speed.   python tools/model_fft2pass.py"""
import numpy as np


def brev(v, bits):
    r = 0
    for i in range(bits):
        r |= ((v >> i) & 1) << (bits - 1 - i)
    return r


def fft32_dif(x, tw, s0, rev):
    """x: (..., 32) registers; tw: None or (..., 5) thread twiddles per stage; five radix-2 DIF stages (fft32.h)."""
    sign = 1.0 if rev else -1.0
    for s in range(s0, 5):
        h = 16 >> s
        y = x.copy()
        for k in range(32):
            if k & h:
                continue
            a, b = x[..., k], x[..., k + h]
            e = (k & (h - 1)) << s
            d = (a - b) * np.exp(sign * 2j * np.pi * e / 32)
            if tw is not None:
                d = d * tw[..., s]
            y[..., k] = a + b
            y[..., k + h] = d
        x = y
    return x


def run(L, L1, rev, rng):
    L2 = L - L1
    N, N1, N2 = 1 << L, 1 << L1, 1 << L2
    T1, T2 = N1 // 32, N2 // 32
    sign = 1.0 if rev else -1.0
    W = lambda n, m: np.exp(sign * 2j * np.pi * (np.asarray(m) % n) / n)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    ws = np.zeros(N, complex)
    # ---- pass 1
    for tile in range(N2 // 16):
        t = np.arange(16 * T1)
        c, u = t & 15, t >> 4
        n2 = tile * 16 + c
        k = np.arange(32)
        reg = x[(u[:, None] + T1 * k[None, :]) * N2 + n2[:, None]]          # rows u + T1 k of column n2
        tw = W(N1, u[:, None] << np.arange(5)[None, :])                       # W_N1^(u << s)
        reg = fft32_dif(reg, tw, 0, rev)
        plane = np.full(N1 * 16, np.nan, complex)
        # writes: ((k*T1)>>5)&1 ? w1 : w0, offset 16*T1*k
        for kk in range(32):
            base = np.where(((kk * T1) >> 5) & 1, (u * 16 + c) ^ 16, u * 16 + c)
            addr = base + 16 * T1 * kk
            assert np.isnan(plane[addr]).all()
            plane[addr] = reg[:, kk]
        assert not np.isnan(plane).any()
        flip = (u & 1) * 16
        reg2 = np.empty_like(reg)
        for kk in range(32):
            addr = np.where(kk & 1, 512 * u + c - flip, 512 * u + c + flip) + 16 * kk
            reg2[:, kk] = plane[addr]
        # check: thread u register kk now holds row 32u + kk of its column
        reg2 = fft32_dif(reg2, None, 10 - L1, rev)
        bu = np.array([brev(int(v), L1 - 5) for v in u])
        for kk in range(32):
            j = brev(kk, 5)
            k1 = (j << (L1 - 5)) | bu
            val = reg2[:, kk] * W(N, n2 * bu) * W(N, (n2 * j) << (L1 - 5))
            ws[tile * (N1 * 16) + k1 * 16 + c] = val
    # ---- pass 2
    out = np.zeros(N, complex)
    for tile in range(N1 // 16):
        t = np.arange(16 * T2)
        ra, ua = t // T2, t % T2
        k = np.arange(32)
        n2 = ua[:, None] + T2 * k[None, :]
        k1 = 16 * tile + ra
        reg = ws[(n2 >> 4) * (N1 * 16) + k1[:, None] * 16 + (n2 & 15)]
        tw = W(N2, ua[:, None] << np.arange(5)[None, :])
        reg = fft32_dif(reg, tw, 0, rev)
        plane = np.full(16 * N2, np.nan, complex)
        for kk in range(32):
            if T2 == 32:
                addr = ra * N2 + np.where(kk & 1, ua ^ ra ^ 16, ua ^ ra) + 32 * kk
            elif T2 == 16:
                addr = ra * N2 + (ua ^ ra) + 16 * ((kk ^ (kk >> 1)) & 1) + 32 * (kk >> 1)
            else:
                b0 = ra * N2 + (ua ^ (ra >> 1)) + 8 * (ra & 1)
                b1 = ra * N2 + (ua ^ (ra >> 1)) + 8 * (1 - (ra & 1))
                addr = np.where(kk & 1, b1, b0) + 16 * (((kk >> 1) ^ (kk >> 2)) & 1) + 32 * (kk >> 2)
            # the kernel's formula must equal the defining swizzle of the slot (rows of 256: the row's mask is its index rotated)
            pos = ua + T2 * kk
            mask = ((ra >> 1) | ((ra & 1) << 3)) if T2 == 8 else ra
            want = ra * N2 + (pos ^ (mask | (((pos >> 5) & 1) << 4)))
            assert (addr == want).all(), (L, L1, kk)
            assert np.isnan(plane[addr]).all()
            plane[addr] = reg[:, kk]
        assert not np.isnan(plane).any()
        rb, ub = t & 15, t >> 4
        rx = (((rb >> 1) | ((rb & 1) << 3)) if T2 == 8 else rb) | ((ub & 1) << 4)
        reg2 = np.empty_like(reg)
        for kk in range(32):
            reg2[:, kk] = plane[rb * N2 + 32 * ub + (kk ^ rx)]
        reg2 = fft32_dif(reg2, None, 10 - L2, rev)
        bub = np.array([brev(int(v), L2 - 5) for v in ub])
        for kk in range(32):
            k2 = (brev(kk, 5) << (L2 - 5)) | bub
            out[k2 * N1 + 16 * tile + rb] = reg2[:, kk]
    ref = np.fft.ifft(x) * N if rev else np.fft.fft(x)
    return np.abs(out - ref).max() / np.abs(ref).max()


def top_stage(x, tw_thread, rev):
    """x: (..., 64) registers = positions u + 32 k of a 2048-point sequence; ONE radix-2 DIF stage over the span 1024: pairs
    (k, k + 32), lower output times W_2048^(u + 32 k) = tw_thread (W_2048^u) x the literal W_64^k."""
    sign = 1.0 if rev else -1.0
    y = x.copy()
    k = np.arange(32)
    a, b = x[..., :32], x[..., 32:]
    y[..., :32] = a + b
    y[..., 32:] = (a - b) * tw_thread[..., None] * np.exp(sign * 2j * np.pi * k / 64)
    return y


def run64(L, L1, rev, rng):
    """the 64-points-per-thread forms (fft_2pass.hip, round 3): a factor of 2048 = 32 threads x 64 points -- one top stage,
    two fft32_dif on the halves, the exchange to 64 consecutive positions, two constants-only fft32_dif -- as pass 2
    (N2 = 2048: N = 2^21 = 1024 x 2048) and as both passes (N = 2^22 = 2048 x 2048)."""
    L2 = L - L1
    N, N1, N2 = 1 << L, 1 << L1, 1 << L2
    sign = 1.0 if rev else -1.0
    W = lambda n, m: np.exp(sign * 2j * np.pi * (np.asarray(m) % n) / n)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    ws = np.zeros(N, complex)
    T1 = N1 // 32
    # ---- pass 1
    for tile in range(N2 // 16):
        if L1 == 11:
            t = np.arange(16 * 32)
            c, u = t & 15, t >> 4
            n2 = tile * 16 + c
            k = np.arange(64)
            reg = x[(u[:, None] + 32 * k[None, :]) * N2 + n2[:, None]]           # rows u + 32 k
            reg = top_stage(reg, W(2048, u), rev)
            tw = W(1024, u[:, None] << np.arange(5)[None, :])
            reg[:, :32] = fft32_dif(reg[:, :32], tw, 0, rev)
            reg[:, 32:] = fft32_dif(reg[:, 32:], tw, 0, rev)
            plane = np.full(2048 * 16, np.nan, complex)
            for kk in range(64):                                                   # row q = 1024 h + u + 32 k', kk = 32 h + k'
                q = 1024 * (kk >> 5) + u + 32 * (kk & 31)
                addr = (q * 16 + c) ^ (((q >> 6) & 1) << 4)
                # the kernel's form: (u*16 + c) ^ (16 * bit) + 512 * k' + 16384 * h, bit = (k' >> 1) & 1
                form = ((u * 16 + c) ^ (16 * (((kk & 31) >> 1) & 1))) + 512 * (kk & 31) + 16384 * (kk >> 5)
                assert (addr == form).all()
                assert np.isnan(plane[addr]).all()
                plane[addr] = reg[:, kk]
            assert not np.isnan(plane).any()
            reg2 = np.empty_like(reg)
            for j in range(64):
                q = 64 * u + j
                addr = (q * 16 + c) ^ (((q >> 6) & 1) << 4)
                flip = 16 * (u & 1)  # the kernel's form: r_even = base + flip, r_odd = base - flip, both + 16 j
                form = 1024 * u + c + 16 * j + np.where(j & 1, -flip, flip)
                assert (addr == form).all()
                reg2[:, j] = plane[addr]
            reg2[:, :32] = fft32_dif(reg2[:, :32], None, 0, rev)
            reg2[:, 32:] = fft32_dif(reg2[:, 32:], None, 0, rev)
            bu = np.array([brev(int(v), 5) for v in u])
            for j in range(64):
                jj = (brev(j & 31, 5) << 1) | (j >> 5)                            # k1 = jj * 32 + bu
                k1 = (jj << 5) | bu
                val = reg2[:, j] * W(N, n2 * bu) * W(N, (n2 * jj) << 5)
                ws[tile * (N1 * 16) + k1 * 16 + c] = val
        else:
            t = np.arange(16 * T1)
            c, u = t & 15, t >> 4
            n2 = tile * 16 + c
            k = np.arange(32)
            reg = x[(u[:, None] + T1 * k[None, :]) * N2 + n2[:, None]]
            tw = W(N1, u[:, None] << np.arange(5)[None, :])
            reg = fft32_dif(reg, tw, 0, rev)
            # (the exchange of the 32-point form is checked by run(); here only its effect: thread u holds rows 32 u + k)
            full = np.empty((16, N1), complex)
            for kk in range(32):
                full[c, u + T1 * kk] = reg[:, kk]
            reg2 = np.stack([full[c, 32 * u + kk] for kk in range(32)], axis=1)
            reg2 = fft32_dif(reg2, None, 10 - L1, rev)
            bu = np.array([brev(int(v), L1 - 5) for v in u])
            for kk in range(32):
                j = brev(kk, 5)
                k1 = (j << (L1 - 5)) | bu
                ws[tile * (N1 * 16) + k1 * 16 + c] = reg2[:, kk] * W(N, n2 * bu) * W(N, (n2 * j) << (L1 - 5))
    # ---- pass 2: rows of 2048, 32 threads x 64 points
    assert L2 == 11
    out = np.zeros(N, complex)
    for tile in range(N1 // 16):
        t = np.arange(16 * 32)
        ra, ua = t // 32, t % 32
        k = np.arange(64)
        n2 = ua[:, None] + 32 * k[None, :]
        k1 = 16 * tile + ra
        reg = ws[(n2 >> 4) * (N1 * 16) + k1[:, None] * 16 + (n2 & 15)]
        reg = top_stage(reg, W(2048, ua), rev)
        tw = W(1024, ua[:, None] << np.arange(5)[None, :])
        reg[:, :32] = fft32_dif(reg[:, :32], tw, 0, rev)
        reg[:, 32:] = fft32_dif(reg[:, 32:], tw, 0, rev)
        plane = np.full(16 * 2048, np.nan, complex)
        for kk in range(64):
            pos = 1024 * (kk >> 5) + ua + 32 * (kk & 31)
            addr = ra * 2048 + (pos ^ (ra | (((pos >> 6) & 1) << 4)))
            form = ra * 2048 + ((ua ^ ra) ^ (16 * (((kk & 31) >> 1) & 1))) + 32 * (kk & 31) + 1024 * (kk >> 5)
            assert (addr == form).all()
            assert np.isnan(plane[addr]).all()
            plane[addr] = reg[:, kk]
        assert not np.isnan(plane).any()
        rb, ub = t & 15, t >> 4
        rx = rb | ((ub & 1) << 4)
        reg2 = np.empty_like(reg)
        for j in range(64):
            pos = 64 * ub + j
            addr = rb * 2048 + (pos ^ (rb | (((pos >> 6) & 1) << 4)))
            form = rb * 2048 + 64 * ub + (j ^ rx)
            assert (addr == form).all()
            reg2[:, j] = plane[addr]
        reg2[:, :32] = fft32_dif(reg2[:, :32], None, 0, rev)
        reg2[:, 32:] = fft32_dif(reg2[:, 32:], None, 0, rev)
        bub = np.array([brev(int(v), 5) for v in ub])
        for j in range(64):
            k2 = (brev(j & 31, 5) << 6) | ((j >> 5) << 5) | bub
            out[k2 * N1 + 16 * tile + rb] = reg2[:, j]
    ref = np.fft.ifft(x) * N if rev else np.fft.fft(x)
    return np.abs(out - ref).max() / np.abs(ref).max()


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for L, L1 in ((16, 8), (17, 8), (18, 9), (19, 9), (20, 10), (17, 9), (19, 10)):
        for rev in (False, True):
            print(f"N = 2^{L} = {1 << L1} x {1 << (L - L1)} {'reverse' if rev else 'forward'}: rel err {run(L, L1, rev, rng):.2e}")
    for L, L1 in ((21, 10), (22, 11)):
        for rev in (False, True):
            print(f"N = 2^{L} = {1 << L1} x {1 << (L - L1)} (64 points per thread) {'reverse' if rev else 'forward'}: rel err {run64(L, L1, rev, rng):.2e}", flush=True)
