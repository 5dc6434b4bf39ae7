#!/usr/bin/env python3
"""(Test infrastructure: uses the oracle, hence it lives under tests/.)  Measured parity margins: normwise error max|got-ref|/max|ref| (SURVEY 8d metric) of every f32 kernel against
the double-precision oracle on seeded random input, next to the 1e-6 bound the tests enforce."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import simpledsp_amd as sd
from oracle import Oracle

o = Oracle()
rng = np.random.default_rng(2024)


def rel(got, ref):
    got = np.asarray(got, dtype=np.complex128 if np.iscomplexobj(ref) else np.float64)
    return float((np.abs(got - ref).max(axis=-1) / np.abs(ref).max(axis=-1)).max())


print("| kernel | case | max normwise error | bound |\n|---|---|---|---|")
for n in (16, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 20, 1 << 21, 1 << 22):
    for radix in (2, 4, 0):
        if radix == 4 and not sd.isPowerOf4(n):
            continue
        if radix == 0 and n != 8192:
            continue  # AUTO: only at 8192 a kernel of its own (mixed radix)
        batch = max(2, min(64, (1 << 16) // n))
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
        errs = []
        for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
            # two-pass sizes: a workspace large enough for their default schedule, the persistent launch (a 256 MiB ring of intermediates)
            plan = sd.FftPlan(n, radix, T, sd.F32, max_batch=max(batch, (1 << 28) // (8 * n)))
            d = torch.from_numpy(x).cuda()
            plan.exec(d)
            torch.cuda.synchronize()
            x128 = x.astype(np.complex128)
            ref = o.fft(x128, radix or 2, rev) if n <= (1 << 20) else (np.fft.ifft(x128, axis=-1) if rev else np.fft.fft(x128, axis=-1))
            errs.append(rel(d.cpu().numpy(), ref))
        print(f"| `{plan.info.kernel.decode()}` | N = {n}, radix {radix}, fwd / rev | {errs[0]:.2e} / {errs[1]:.2e} | 1e-6 |")

# double precision (round 3): against the oracle in units of the reference's own bound 4 N eps (testFFT.cpp:37)
for n in (64, 1024, 4096, 8192, 16384, 1 << 15, 1 << 16, 1 << 18, 1 << 20):
    for radix in (2, 4):
        if radix == 4 and not sd.isPowerOf4(n):
            continue
        batch = max(2, min(16, (1 << 16) // n))
        x = rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))
        errs = []
        for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
            plan = sd.FftPlan(n, radix, T, sd.F64, max_batch=max(batch, (1 << 28) // (16 * n)))
            d = torch.from_numpy(x).cuda()
            plan.exec(d)
            torch.cuda.synchronize()
            errs.append(rel(d.cpu().numpy(), o.fft(x, radix, rev)))
        tol = 4 * n * np.finfo(np.float64).eps
        print(f"| `{plan.info.kernel.decode()}` (f64) | N = {n}, radix {radix}, fwd / rev | {errs[0]:.2e} / {errs[1]:.2e} (= {errs[0] / tol:.3f} / {errs[1] / tol:.3f} of the bound) | 4 N eps = {tol:.1e} |")

# cascaded biquads, BASELINE config-4 filter, f32
x = rng.standard_normal((64, 4096)).astype(np.float32)
bank = sd.casc_2o_iir(4, 64, sd.F32, sd.IIR_GENERIC)
bank.set_lp_coeff(10e3, 100e3)
d = torch.from_numpy(x).cuda()
bank.process(d)
torch.cuda.synchronize()
fo = o.iir(4)
want = []
for c in range(64):
    f = o.iir(4)
    f.set_lp_coeff(10e3, 100e3)
    want.append(f.process(x[c].astype(np.float64)))
print(f"| `{bank.kernel_name(d)}` | 4-section LP, 4096 samples, f32 | {rel(d.cpu().numpy(), np.array(want)):.2e} | 1e-6 (f64: bit-exact) |")

# the same filter and a low cutoff (f0/fs = 0.005) in pure f32 and in the mixed mode (float samples, double recurrence)
for f0 in (10e3, 500.0):
    for prec, name in ((sd.F32, "f32"), (sd.F32_F64STATE, "float samples + double recurrence")):
        bank = sd.casc_2o_iir(4, 64, prec, sd.IIR_GENERIC)
        bank.set_lp_coeff(f0, 100e3)
        d = torch.from_numpy(x).cuda()
        bank.process(d)
        torch.cuda.synchronize()
        want = []
        for c in range(64):
            f = o.iir(4)
            f.set_lp_coeff(f0, 100e3)
            want.append(f.process(x[c].astype(np.float64)))
        print(f"| `{bank.kernel_name(d)}` | 4-section LP f0/fs = {f0 / 100e3:g}, 4096 samples, {name} | {rel(d.cpu().numpy(), np.array(want)):.2e} | "
              f"{'1e-6 (BASELINE filter only)' if prec == sd.F32 else '1.2e-7'} |")

# FIR, f32
for taps in (16, 32, 64):
    fb = sd.fir_filter(taps, 64, sd.F32)
    fb.set_lp_coeff(10e3, 100e3)
    d = torch.from_numpy(x).cuda()
    fb.process(d)
    torch.cuda.synchronize()
    h32 = fb.m_coeff.astype(np.float32).astype(np.float64)
    want = np.array([o.fir_process(h32, x[c].astype(np.float64))[0] for c in range(64)])
    print(f"| `sdsp_fir_kernel` | {taps}-tap LP, 4096 samples, f32 | {rel(d.cpu().numpy(), want):.2e} | 1e-6 (f64: bit-exact) |")

# fused convolution
for n, radix in ((4096, 4), (1024, 2)):
    xb = (rng.standard_normal((8, n)) + 1j * rng.standard_normal((8, n))).astype(np.complex64)
    hh = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    plan = sd.FftPlan(n, radix, sd.forward_fft, sd.F32, max_batch=8)
    d = torch.from_numpy(xb).cuda()
    plan.convolve(d, torch.from_numpy(hh).cuda())
    torch.cuda.synchronize()
    ref = o.fft(o.fft(xb.astype(np.complex128), radix) * hh.astype(np.complex128), radix, True)
    print(f"| fused convolution | N = {n}, radix {radix} | {rel(d.cpu().numpy(), ref):.2e} | 2e-6 (two transforms) |")
