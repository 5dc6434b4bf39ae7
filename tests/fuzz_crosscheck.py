#!/usr/bin/env python3
"""(Test infrastructure: uses the oracle, hence it lives under tests/.)  Randomised cross-check of the C-ABI paths against numpy / the oracle: shapes, strides, offsets, variants.
   tests/fuzz_crosscheck.py [seed] [cases]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import simpledsp_amd as sd
from oracle import Oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
o = Oracle()
bad = 0


def rel(got, ref):
    return float(np.abs(np.asarray(got) - ref).max() / max(np.abs(ref).max(), 1e-300))


default_piece = sd.get_launch_piece_bytes()
for case in range(cases):
    kind = rng.choice(["fft", "fft", "iir", "fir", "conv", "rfft"])
    # launch pieces (sdsp_hip_set_launch_piece_bytes): small pieces so that the batches of these cases are split too
    sd.set_launch_piece_bytes(int(rng.choice([default_piece, default_piece, 0, 1 << 20, 3 << 19, 5 << 20])))
    try:
        if kind in ("fft", "conv", "rfft"):
            log2n = int(rng.integers(1, 23)) if kind == "fft" else int(rng.integers(4, 16))  # up to 2^22 (two-pass since round 3)
            n = 1 << log2n
            radix = 4 if (log2n % 2 == 0 and rng.random() < 0.5) else 2
            f64 = rng.random() < 0.3 and n <= ((1 << 20) if kind == "fft" else 16384)  # fft: .. 2^20 (two-pass kernels in double); conv / rfft in
            # double: .. 16384 (round 3: inside the f64 registers-resident kernel from 4096)
            if f64 and kind == "rfft" and radix == 4 and n > 4096:
                radix = 2  # radix-4 real-input plans in double stop at n_real = 8192
            batch = int(rng.integers(1, max(3, min(300, (1 << 18) // n))))
            rev = bool(rng.integers(0, 2))
            cdt = np.complex128 if f64 else np.complex64
            tol = (8 * n * 2.3e-16) if f64 else 1e-6
            x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(cdt)
            if kind == "fft":
                plan = sd.FftPlan(n, radix, sd.reverse_fft if rev else sd.forward_fft, sd.F64 if f64 else sd.F32,
                                  max_batch=int(rng.integers(1, batch + 1)))
                if rng.random() < 0.3:  # every kernel variant a plan of this shape can reach (out-of-range ones fall back)
                    plan.set_variant(int(rng.integers(0, 24)) if (n, radix, f64) == (4096, 4, False) else int(rng.integers(0, 10)))
                d = torch.from_numpy(x).cuda()
                plan.exec(d)
                torch.cuda.synchronize()
                ref = np.fft.ifft(x.astype(np.complex128), axis=-1) if rev else np.fft.fft(x.astype(np.complex128), axis=-1)
                err = rel(d.cpu().numpy(), ref)
                desc = f"fft n={n} r{radix} {'f64' if f64 else 'f32'} batch={batch} rev={rev} var={plan._variant if hasattr(plan, '_variant') else '?'}"
            elif kind == "conv":
                h = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
                plan = sd.FftPlan(n, radix, sd.forward_fft, sd.F64 if f64 else sd.F32, max_batch=batch)
                cvar = int(rng.choice([0, 0, 1, 2]))  # fused kernel of the size / three launches / the register-pass fusion
                plan.set_variant(cvar)
                d = torch.from_numpy(x).cuda()
                plan.convolve(d, torch.from_numpy(h).cuda())
                torch.cuda.synchronize()
                ref = np.fft.ifft(np.fft.fft(x.astype(np.complex128), axis=-1) * h.astype(np.complex128), axis=-1)
                err, tol = rel(d.cpu().numpy(), ref), (16 * n * 2.3e-16 if f64 else 2e-6)
                desc = f"conv n={n} r{radix} {'f64' if f64 else 'f32'} batch={batch} var={cvar}"
            else:
                n_real = 2 * n
                xr = rng.standard_normal((batch, n_real)).astype(np.float64 if f64 else np.float32)
                prec = sd.F64 if f64 else sd.F32
                fwd = sd.RfftPlan(n_real, radix, sd.forward_fft, max_batch=batch, precision=prec)
                inv = sd.RfftPlan(n_real, radix, sd.reverse_fft, max_batch=batch, precision=prec)
                rvar = int(rng.choice([0, 0, 1, 2]))  # the size's default kernel / the register-pass family's two cache policies
                if n > (8192 if f64 else 16384):
                    rvar = 0  # n_real = 65536 (double: 32768) exists in the registers-resident kernel only
                if f64:  # double: one alternate (the register-pass family's MODE 1 / 2), where the registers-resident kernel is the default
                    rvar = min(rvar, 1) if (radix == 2 and n in (4096, 8192)) else 0
                fwd.set_variant(rvar)
                inv.set_variant(int(rng.choice([0, rvar])))
                d = torch.from_numpy(xr).cuda()
                fwd.exec(d)
                torch.cuda.synchronize()
                spec = d.cpu().numpy().view(cdt).reshape(batch, n)
                ref = np.fft.rfft(xr.astype(np.float64), axis=-1)
                got = spec.astype(np.complex128).copy()
                e0 = np.abs(got[:, 1:] - ref[:, 1:n]).max() / np.abs(ref).max()
                e1 = max(np.abs(spec[:, 0].real - ref[:, 0].real).max(), np.abs(spec[:, 0].imag - ref[:, n].real).max()) / np.abs(ref).max()
                inv.exec(d)
                torch.cuda.synchronize()
                e2 = rel(d.cpu().numpy(), xr.astype(np.float64))
                err, tol = max(e0, e1, e2 / 4), (8 * n_real * 2.3e-16 if f64 else 1e-6)
                desc = f"rfft n_real={n_real} r{radix} {'f64' if f64 else 'f32'} batch={batch} var={rvar}"
        elif kind == "iir":
            m = int(rng.choice([2, 4, 6, 8]))
            mode = int(rng.integers(0, 3))  # 0 f32, 1 f64, 2 float samples + double recurrence (SDSP_HIP_F32_F64STATE)
            f64, mixed = mode == 1, mode == 2
            channels = int(rng.integers(1, 400))
            samples = int(rng.integers(1, 700))
            pad = int(rng.choice([0, 0, 1, 3, 4, 16]))
            off = int(rng.integers(0, pad + 1))
            ftype = int(rng.integers(1, 5))
            spec_kind = ftype if (ftype < 4 and rng.random() < 0.5) else sd.IIR_GENERIC
            bank = sd.casc_2o_iir(m, channels, (sd.F32, sd.F64, sd.F32_F64STATE)[mode], spec_kind)
            # f32 recursions lose accuracy as the poles approach z = 1 (1.3e-4 at f0/fs = 200/39000): the pure
            # f32 cases stay in the BASELINE filter's neighbourhood; the f64 cases (bit-exact) and the mixed mode (float
            # samples, double recurrence: 1e-6 everywhere) roam over f0 in [500, 20k]
            args = (float(rng.uniform(8e3, 20e3) if mode == 0 else rng.uniform(500, 20e3)), 100e3)
            {1: lambda: bank.set_lp_coeff(*args), 2: lambda: bank.set_hp_coeff(*args), 3: lambda: bank.set_bp_coeff(*args, 1.3),
             4: lambda: bank.set_bs_coeff(*args, 1.3)}[ftype]()
            bank.set_variant(int(rng.integers(0, 4)))
            dt = np.float64 if f64 else np.float32
            x = rng.standard_normal((channels, samples + pad)).astype(dt)
            cut = int(rng.integers(0, samples + 1))
            wire = rng.random() < 0.3 and (f64 or channels % 2 == 0)  # sample-major "wire" layout
            if wire:
                bank.set_variant(int(rng.integers(0, 3)))
                dw = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()  # (samples + pad, channels)
                if cut:
                    bank.process_interleaved(dw, samples=cut, offset=off)
                if samples - cut:
                    bank.process_interleaved(dw, samples=samples - cut, offset=off + cut)
                torch.cuda.synchronize()
                got = dw.cpu().numpy().T
            else:
                d = torch.from_numpy(x).cuda()
                if cut:
                    bank.process(d, samples=cut, offset=off)
                if samples - cut:
                    bank.process(d, samples=samples - cut, offset=off + cut)
                torch.cuda.synchronize()
                got = d.cpu().numpy()
            err = 0.0
            for c in sorted({0, channels - 1, int(rng.integers(0, channels))}):
                fo = o.iir(m)
                fo.set_design(bank.m_a_coeff, bank.m_b_coeff, bank.m_gain, ftype)
                want = fo.process(x[c, off:off + samples].astype(np.float64), spec_kind)
                if f64:
                    err = max(err, 0.0 if np.array_equal(got[c, off:off + samples], want) else 1.0)
                else:
                    err = max(err, rel(got[c, off:off + samples], want))
                # untouched padding stays untouched
                if not np.array_equal(got[c, :off], x[c, :off]) or not np.array_equal(got[c, off + samples:], x[c, off + samples:]):
                    err = 9.0
            # pure f32: 8 sections accumulate more rounding than the 4 of the BASELINE filter (1e-6); mixed: one float rounding
            tol = 0.5 if f64 else 1e-6 if mixed else 3e-6
            desc = f"iir{' wire' if wire else ''} m={m} {('f32', 'f64', 'f32+f64state')[mode]} ch={channels} n={samples} pad={pad} off={off} type={ftype} kind={spec_kind} var={bank._variant}"
        else:
            taps = int(rng.integers(1, 200))
            f64 = rng.random() < 0.5
            channels = int(rng.integers(1, 300))
            samples = int(rng.integers(1, 6000))
            pad = int(rng.choice([0, 0, 1, 3, 4]))
            off = int(rng.integers(0, pad + 1))
            bank = sd.fir_filter(taps, channels, sd.F64 if f64 else sd.F32)
            h = rng.standard_normal(taps) / np.sqrt(taps)
            bank.set_coeff(h)
            bank.set_variant(int(rng.integers(0, 4)))
            dt = np.float64 if f64 else np.float32
            x = rng.standard_normal((channels, samples + pad)).astype(dt)
            d = torch.from_numpy(x).cuda()
            cut = int(rng.integers(0, samples + 1))
            if cut:
                bank.process(d, samples=cut, offset=off)
            if samples - cut:
                bank.process(d, samples=samples - cut, offset=off + cut)
            torch.cuda.synchronize()
            got = d.cpu().numpy()
            err = 0.0
            hh = h if f64 else h.astype(np.float32).astype(np.float64)
            for c in sorted({0, channels - 1, int(rng.integers(0, channels))}):
                want = o.fir_process(hh, x[c, off:off + samples].astype(np.float64))[0]
                if f64:
                    err = max(err, 0.0 if np.array_equal(got[c, off:off + samples], want) else 1.0)
                else:
                    err = max(err, rel(got[c, off:off + samples], want))
                if not np.array_equal(got[c, :off], x[c, :off]) or not np.array_equal(got[c, off + samples:], x[c, off + samples:]):
                    err = 9.0
            tol = 0.5 if f64 else 2e-6
            desc = f"fir taps={taps} {'f64' if f64 else 'f32'} ch={channels} n={samples} pad={pad} off={off} var={bank._variant}"
        if not (err < tol):
            bad += 1
            print(f"FAIL case {case}: {desc}: err {err:.3e} (tol {tol:.1e})", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"EXC case {case} ({kind}): {type(e).__name__}: {e}", flush=True)
print(f"seed {seed}: {cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
