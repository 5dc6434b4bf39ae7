"""Shared pytest plumbing.

`-m "not gpu"`: oracle vs golden vectors / the reference's own known-answer tests, host logic,
C-ABI symbol checks -- no GPU needed.  `-m gpu`: the parity tests proper, through the C-ABI, on
a real MI355X.  Nothing here reads /root/reference at run time.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def fft_golden():
    return np.load(GOLDEN / "fft_golden.npz")


@pytest.fixture(scope="session")
def iir_golden():
    return np.load(GOLDEN / "iir_golden.npz")


def read_impulse_csv(path):
    """type,fs,f0,Q,n,v0..v(n-1) on one line -- the format testIIR.cpp:7-28 parses."""
    v = np.array(Path(path).read_text().strip().split(","), dtype=np.float64)
    n = int(v[4])
    return int(v[0]), float(v[1]), float(v[2]), float(v[3]), v[5:5 + n]


def impulse_csvs():
    return sorted((GOLDEN / "impulse_response").glob("*.csv"))


def design(f, ftype, f0, fs, q, gain_in=1.0):
    if ftype == 1:
        f.set_lp_coeff(f0, fs, gain_in)
    elif ftype == 2:
        f.set_hp_coeff(f0, fs, gain_in)
    elif ftype == 3:
        f.set_bp_coeff(f0, fs, q, gain_in)
    elif ftype == 4:  # band-stop: not in the reference (README TODO), see test_band_stop_*
        f.set_bs_coeff(f0, fs, q, gain_in)
    else:
        raise RuntimeError("Unknown filter type")


def band_edges(f0, fs, q):
    """-3 dB edges (Hz) of the set_bp_coeff / set_bs_coeff parameterisation: width f0/q, centre
    angle e0 with cos(e0) = cos((e1+e2)/2) / cos((e2-e1)/2) -- the relation the reference's
    test_data/findIIRCutoffFreq.m solves numerically for its Octave band-pass fixtures."""
    e0 = 2 * np.pi * f0 / fs
    d = e0 / q
    mid = np.arccos(np.cos(e0) * np.cos(d / 2))
    return (mid - d / 2) * fs / (2 * np.pi), (mid + d / 2) * fs / (2 * np.pi)


def scipy_band_stop_sos(m, f0, fs, q):
    """Third opinion for the band-stop design: Butterworth order m between band_edges."""
    import scipy.signal
    return scipy.signal.butter(m, band_edges(f0, fs, q), "bandstop", fs=fs, output="sos")


BAND_STOP_CASES = [(200.0, 39e3, 1.4), (2000.0, 39e3, 0.8), (15000.0, 39e3, 2.0), (10e3, 100e3, 1.1)]


def rel_max_err(got, ref):
    """SURVEY 8(d) metric: max_k |got-ref| / max_k |ref| per transform (last axis)."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    num = np.abs(got - ref).max(axis=-1)
    den = np.abs(ref).max(axis=-1)
    return float(np.max(num / np.where(den == 0, 1.0, den)))
