"""The CPU oracle's FIR filter (README.md:16 TODO in the reference: NO reference code exists, so parity
with the reference is UNPINNED for this row).  It is pinned to scipy instead (SURVEY 8(c) "independent
third opinions"): the design to scipy.signal.firwin, the filter to scipy.signal.lfilter."""
import numpy as np
import pytest

FIR_CASES = [  # taps, filter_type, f0, fs, q, firwin kwargs
    (1, 1, 10e3, 100e3, 0.0, dict(cutoff=10e3)),
    (2, 1, 10e3, 100e3, 0.0, dict(cutoff=10e3)),
    (31, 1, 10e3, 100e3, 0.0, dict(cutoff=10e3)),
    (32, 1, 2e3, 39e3, 0.0, dict(cutoff=2e3)),
    (33, 2, 10e3, 100e3, 0.0, dict(cutoff=10e3, pass_zero=False)),
    (64, 3, 10e3, 100e3, 1.1, dict(cutoff=[10e3 - 10e3 / 2.2, 10e3 + 10e3 / 2.2], pass_zero=False)),
    (65, 4, 10e3, 100e3, 1.1, dict(cutoff=[10e3 - 10e3 / 2.2, 10e3 + 10e3 / 2.2])),
    (257, 1, 200.0, 39e3, 0.0, dict(cutoff=200.0)),
]


@pytest.mark.parametrize("taps,ftype,f0,fs,q,kw", FIR_CASES)
def test_design_matches_firwin(oracle, taps, ftype, f0, fs, q, kw):
    import scipy.signal
    h = oracle.fir_design(taps, ftype, f0, fs, q)
    assert np.abs(h - scipy.signal.firwin(taps, fs=fs, **kw)).max() < 1e-15
    assert np.abs(oracle.fir_design(taps, ftype, f0, fs, q, 2.5) - 2.5 * h).max() < 1e-15


def test_design_rejects_what_firwin_rejects(oracle):
    for taps, ftype in ((32, 2), (64, 4), (0, 1)):
        with pytest.raises(ValueError):
            oracle.fir_design(taps, ftype, 10e3, 100e3, 1.1)


@pytest.mark.parametrize("taps", [1, 2, 5, 16, 17, 31, 32, 33, 100])
def test_filter_matches_lfilter_and_streams(oracle, taps):
    import scipy.signal
    rng = np.random.default_rng(taps)
    h = rng.standard_normal(taps)
    x = rng.standard_normal(700)
    y, mem = oracle.fir_process(h, x)
    assert np.abs(y - scipy.signal.lfilter(h, 1.0, x)).max() < 1e-13
    assert np.array_equal(mem, x[::-1][: taps - 1])
    # block by block == one long call, bit for bit (testIIR.cpp:61-75 semantics)
    out, m = [], None
    for a, b in ((0, 1), (1, 33), (33, 300), (300, 700)):
        yb, m = oracle.fir_process(h, x[a:b], m)
        out.append(yb)
    assert np.array_equal(np.concatenate(out), y)
    # impulse response == the taps
    imp = np.zeros(taps + 3)
    imp[0] = 1.0
    assert np.array_equal(oracle.fir_process(h, imp)[0][:taps], h)
