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
    else:
        raise RuntimeError("Unknown filter type")


def rel_max_err(got, ref):
    """SURVEY 8(d) metric: max_k |got-ref| / max_k |ref| per transform (last axis)."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    num = np.abs(got - ref).max(axis=-1)
    den = np.abs(ref).max(axis=-1)
    return float(np.max(num / np.where(den == 0, 1.0, den)))
