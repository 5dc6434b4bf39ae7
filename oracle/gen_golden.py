#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (oracle/_ref/libsdsp_ref.so).

TEST INFRASTRUCTURE ONLY.  Run in the build container, where /root/reference exists:

    make -C oracle && python oracle/gen_golden.py

Writes data only (inputs + the reference's outputs):
  tests/golden/fft_golden.npz      inputs/outputs of sdsp::fft_radix2 / fft_radix4 (fft.h:258-360)
  tests/golden/iir_golden.npz      outputs/coefficients/state of sdsp::casc_2o_iir* (casc_2o_iir.h)
  tests/golden/impulse_response/   the 9 Octave CSV fixtures the reference's own tests read
                                   (test_data/impulse_response/*.csv, testIIR.cpp:34) -- data files
The input recipes restate the reference's test inputs (testFFT.cpp / testIIR.cpp lines cited inline).
"""
from __future__ import annotations

import shutil
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import Reference  # noqa: E402

GOLD = ROOT / "tests" / "golden"
REF_CSV = Path("/root/reference/test_data/impulse_response")
SEED = 0x5D5B


def f32_representable(rng, shape):
    """N(0,1) values exactly representable in fp32, so fp32 and fp64 paths see identical inputs."""
    return rng.standard_normal(shape).astype(np.float32).astype(np.float64)


def gen_fft(ref: Reference) -> dict:
    g: dict = {}

    def both(tag, x, radices=(2, 4), reverse=(False, True)):
        g[f"{tag}__in"] = x
        for r in radices:
            for rev in reverse:
                g[f"{tag}__r{r}_{'rev' if rev else 'fwd'}"] = ref.fft(x, r, rev)

    # testFFT.cpp:19-29 / :129-139: N=64 real cosine at bin 7, and its analytic spectrum
    N, n = 64, 7
    i = np.arange(N, dtype=np.float64)
    s = np.cos(n * 2 * np.pi * i / N).astype(np.complex128)
    S = np.zeros(N, np.complex128)
    S[n] = N / 2
    S[N - n] = N / 2
    both("cos64", s)
    both("spec64", S)
    # testFFT.cpp:52-56: same cosine shifted by +90 degrees
    s2 = np.cos(n * 2 * np.pi * i / N + (np.pi / 2.0)).astype(np.complex128)
    both("cos64_shift90", s2)

    # testFFT.cpp:72-101: linearity inputs, N=256, 1 kHz / 500 Hz sines at fs=8 kHz
    N = 256
    i = np.arange(N, dtype=np.float64)
    x1 = np.sin(2.0 * np.pi * 1000.0 * (1.0 / 8000.0) * i).astype(np.complex128)
    x2 = np.sin(2.0 * np.pi * 500.0 * (1.0 / 8000.0) * i).astype(np.complex128)
    both("lin256_x1", x1, reverse=(False,))
    both("lin256_x2", x2, reverse=(False,))
    both("lin256_sum", 1.5 * x1 + 2.5 * x2, reverse=(False,))

    # testFFT.cpp:239: the N=1024 benchmark vector (BASELINE config 1)
    b = np.zeros(1024, np.complex128)
    b[:8] = [0.3535, 0.3535, 0.6464, 1.0607, 0.3535, -1.0607, -1.3535, -0.3535]
    both("bench1024", b)

    # seeded random complex, fp32-representable, every size the shim instantiates
    rng = np.random.default_rng(SEED)
    for N in Reference.SIZES:
        batch = 2 if N >= 1024 else 3
        x = f32_representable(rng, (batch, N)) + 1j * f32_representable(rng, (batch, N))
        radices = (2, 4) if ref.is_power_of_4(N) else (2,)
        both(f"rand{N}", x, radices=radices)

    # tables: fft.h:197-256
    g["wcoeffs64_fwd"] = ref.wcoeffs(64, False)
    g["wcoeffs64_rev"] = ref.wcoeffs(64, True)
    g["wcoeffs4096_lastrow_fwd"] = ref.wcoeffs(4096, False)[-1]
    g["wcoeffs1024_row4_fwd"] = ref.wcoeffs(1024, False)[4]
    for N in (16, 64, 4096):
        g[f"swap{N}_base2"] = ref.swap_lookup(N, 2)
        g[f"swap{N}_base4"] = ref.swap_lookup(N, 4)
    g["swap128_base2"] = ref.swap_lookup(128, 2)
    return g


def read_csv(path: Path):
    """format: type,fs,f0,Q,n,v0..v(n-1) on one line (testIIR.cpp:7-28)."""
    v = np.array(path.read_text().strip().split(","), dtype=np.float64)
    return int(v[0]), v[1], v[2], v[3], v[5:5 + int(v[4])]


def design(f, ftype, f0, fs, q, gain=1.0):
    if ftype == 1:
        f.set_lp_coeff(f0, fs, gain)
    elif ftype == 2:
        f.set_hp_coeff(f0, fs, gain)
    else:
        f.set_bp_coeff(f0, fs, q, gain)


def gen_iir(ref: Reference) -> dict:
    g: dict = {}
    names = []
    # testIIR.cpp:32-75 (+ :223-251, :304-332, :385-413): impulse response per CSV parameter set
    for csv in sorted(REF_CSV.glob("*.csv")):
        ftype, fs, f0, q, imp = read_csv(csv)
        tag = csv.stem
        names.append(tag)
        g[f"{tag}__params"] = np.array([ftype, fs, f0, q])
        x = np.zeros(imp.size)
        x[0] = 1.0
        f = ref.iir(4, 0)
        design(f, ftype, f0, fs, q)
        a, b, gain = f.coeffs()
        g[f"{tag}__a"], g[f"{tag}__b"], g[f"{tag}__gain"] = a, b, np.array(gain)
        g[f"{tag}__generic"] = f.process(x)
        mem, pos = f.state()
        g[f"{tag}__generic_mem"], g[f"{tag}__generic_pos"] = mem, np.array(pos)
        fsp = ref.iir(4, ftype)
        design(fsp, ftype, f0, fs, q)
        g[f"{tag}__spec"] = fsp.process(x)
    g["csv_names"] = np.array(names)

    # testIIR.cpp:79-171 gain tests and :465-559 benchmark filters: fs=100k, f0=10k, Q=1.1
    fs, f0, q = 100e3, 10e3, 1.1
    rng = np.random.default_rng(SEED + 2)
    xr = f32_representable(rng, 4096)
    g["rand4096__in"] = xr
    for ftype, nm in ((1, "lp"), (2, "hp"), (3, "bp")):
        for gain_in in (1.0, 2.0):
            x = np.zeros(1024)
            x[0] = 1.0
            f = ref.iir(4, 0)
            design(f, ftype, f0, fs, q, gain_in)
            g[f"gain_{nm}_{gain_in:g}__generic"] = f.process(x)
            a, b, gn = f.coeffs()
            g[f"gain_{nm}_{gain_in:g}__a"], g[f"gain_{nm}_{gain_in:g}__b"] = a, b
            g[f"gain_{nm}_{gain_in:g}__gain"] = np.array(gn)
            fsp = ref.iir(4, ftype)
            design(fsp, ftype, f0, fs, q, gain_in)
            g[f"gain_{nm}_{gain_in:g}__spec"] = fsp.process(x)
        # 4096-sample impulse (benchmark body) and a seeded random channel
        x = np.zeros(4096)
        x[0] = 1.0
        for kind, kn in ((0, "generic"), (ftype, "spec")):
            f = ref.iir(4, kind)
            design(f, ftype, f0, fs, q)
            g[f"bench4096_{nm}__{kn}"] = f.process(x)
            f = ref.iir(4, kind)
            design(f, ftype, f0, fs, q)
            g[f"rand4096_{nm}__{kn}"] = f.process(xr)
        # preload: testIIR.cpp:173-218
        f = ref.iir(4, 0)
        design(f, ftype, f0, fs, q)
        f.preload_filter(10.0)
        mem, pos = f.state()
        g[f"preload_{nm}__mem"] = mem
        g[f"preload_{nm}__out"] = f.process(np.full(1024, 10.0))

    # other section counts (m_t = 2, 6, 8), random input, generic + specialised
    xs = f32_representable(rng, 512)
    g["rand512__in"] = xs
    for m in (2, 6, 8):
        for ftype, nm in ((1, "lp"), (2, "hp"), (3, "bp")):
            for kind, kn in ((0, "generic"), (ftype, "spec")):
                f = ref.iir(m, kind)
                design(f, ftype, 3e3, 48e3, 0.9)
                g[f"rand512_m{m}_{nm}__{kn}"] = f.process(xs)
    return g


def main():
    ref = Reference()
    GOLD.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLD / "fft_golden.npz", **gen_fft(ref))
    np.savez_compressed(GOLD / "iir_golden.npz", **gen_iir(ref))
    dst = GOLD / "impulse_response"
    dst.mkdir(exist_ok=True)
    for csv in sorted(REF_CSV.glob("*.csv")):
        shutil.copyfile(csv, dst / csv.name)
    for p in sorted(GOLD.rglob("*")):
        if p.is_file():
            print(f"{p.relative_to(ROOT)}  {p.stat().st_size} B")


if __name__ == "__main__":
    main()
