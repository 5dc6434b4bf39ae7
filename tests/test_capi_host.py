"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/sdsp_hip.h declares; the host-side (cold) functions -- size helpers, run-time twiddle
precompute, coefficient design, preload -- agree with the oracle.  No compute call needs a GPU
here; the compute entry points must FAIL LOUDLY without one (no CPU fallback)."""
import ctypes as C
import re

import numpy as np
import pytest

from conftest import ROOT, design

import simpledsp_amd as sd
from simpledsp_amd import _lib as L


def _declared_symbols():
    text = (ROOT / "include" / "sdsp_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdsp_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = sd.load()
    names = _declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in sdsp_hip.h but not exported"
    assert set(names) == set(L.SIGNATURES), "python binding and header disagree"
    assert b"gfx950" in lib.sdsp_hip_version()


def test_size_helpers_match_oracle(oracle):
    for v in [0, 1, 2, 3, 4, 5, 8, 16, 63, 64, 96, 1024, 2048, 4096, 1 << 20]:
        assert sd.isPowerOf2(v) == oracle.is_power_of_2(v)
        assert sd.isPowerOf4(v) == oracle.is_power_of_4(v)
        if v:
            assert sd.log2(v) == oracle.log2(v) and sd.log4(v) == oracle.log4(v)
    for n, base in [(64, 2), (64, 4), (4096, 4), (4096, 2), (128, 2), (16, 4)]:
        for x in range(n):
            assert sd.digit_reverse(n, base, x) == oracle.digit_reverse(n, base, x)
        assert np.array_equal(sd.calc_swap_lookup(n, base), oracle.swap_lookup(n, base))


def test_runtime_twiddles_match_reference_table(oracle, fft_golden):
    # a4: run-time precompute vs the reference's compile-time table (fixtures from the real reference)
    assert np.array_equal(sd.calc_wCoeffs(64, sd.forward_fft), fft_golden["wcoeffs64_fwd"])
    assert np.array_equal(sd.calc_wCoeffs(64, sd.reverse_fft), fft_golden["wcoeffs64_rev"])
    assert np.array_equal(sd.calc_twiddles(4096), fft_golden["wcoeffs4096_lastrow_fwd"])
    assert np.array_equal(sd.calc_wCoeffs(1024)[4], fft_golden["wcoeffs1024_row4_fwd"])
    for n in (2, 4, 8, 256, 2048):
        assert np.array_equal(sd.calc_wCoeffs(n), oracle.wcoeffs(n))
    w = sd.calc_twiddles(1 << 20)  # beyond what the reference can compile: <= 1 ulp from exact
    j = np.arange(1 << 20)
    assert np.abs(w - np.exp(-2j * np.pi * j / (1 << 20))).max() < 2e-15  # numpy arg rounding
    assert w[1 << 18] == -1j and w[1 << 19] == -1 and w[0] == 1  # exact mirror symmetry


@pytest.mark.parametrize("m", [2, 4, 6, 8])
def test_coefficient_design_matches_oracle(oracle, m):
    for ftype, args in ((1, (10e3, 100e3)), (2, (200.0, 39e3)), (3, (2e3, 39e3, 0.8))):
        f = sd.casc_2o_iir(m)
        fo = oracle.iir(m)
        q = args[2] if ftype == 3 else 0.0
        design(f, ftype, args[0], args[1], q, 1.7)
        design(fo, ftype, args[0], args[1], q, 1.7)
        assert np.array_equal(f.m_a_coeff, fo.a) and np.array_equal(f.m_b_coeff, fo.b)
        assert f.m_gain == fo.gain and f.m_f_type == fo.f_type


@pytest.mark.parametrize("m", [2, 4, 8])
def test_band_stop_design_matches_oracle_and_scipy(oracle, m):
    """sdsp_hip_iir_design_bs (README.md:15 TODO, no reference code): the C++ product version against the
    independent C99 oracle version (rounding-level agreement: different complex-arithmetic libraries) and
    against scipy's Butterworth band-stop on the impulse response."""
    import scipy.signal
    from conftest import BAND_STOP_CASES, scipy_band_stop_sos
    for f0, fs, q in BAND_STOP_CASES:
        f = sd.casc_2o_iir(m)
        f.set_bs_coeff(f0, fs, q, 1.7)
        fo = oracle.iir(m)
        fo.set_bs_coeff(f0, fs, q, 1.7)
        assert f.m_f_type == fo.f_type == 4
        assert np.array_equal(f.m_b_coeff, fo.b)
        assert np.abs(f.m_a_coeff - fo.a).max() < 1e-14 and abs(f.m_gain / fo.gain - 1) < 1e-11
        sos = np.hstack([f.m_b_coeff, f.m_a_coeff])
        sos[0, :3] *= f.m_gain / 1.7
        x = np.zeros(1000)
        x[0] = 1.0
        assert np.abs(scipy.signal.sosfilt(sos, x) - scipy.signal.sosfilt(scipy_band_stop_sos(m, f0, fs, q), x)).max() < 1e-12
    # preload of a band-stop propagates DC through the sections like low_pass
    lib = sd.load()
    f = sd.casc_2o_iir(4)
    f.set_bs_coeff(10e3, 100e3, 1.1)
    mem = np.zeros((5, 3))
    L.check(lib.sdsp_hip_iir_preload(4, f.m_f_type, f.m_a_coeff.ctypes.data, f.m_b_coeff.ctypes.data, f.m_gain, 10.0,
                                     mem.ctypes.data))
    fo = oracle.iir(4)
    fo.set_bs_coeff(10e3, 100e3, 1.1)
    fo.preload_filter(10.0)
    assert np.abs(mem - fo.mem).max() < 1e-10 and abs(mem[4, 0] - 10.0) < 1e-9


def test_fir_design_matches_oracle_and_firwin(oracle):
    """sdsp_hip_fir_design (README.md:16 TODO, no reference code) against the oracle's independent C
    version and scipy.signal.firwin; error codes for what firwin rejects."""
    import scipy.signal
    from test_oracle_fir import FIR_CASES
    lib = sd.load()
    for taps, ftype, f0, fs, q, kw in FIR_CASES:
        f = sd.fir_filter(taps)
        {1: f.set_lp_coeff, 2: f.set_hp_coeff}.get(ftype, lambda a, b, g=1.0: None)(f0, fs)
        if ftype == 3:
            f.set_bp_coeff(f0, fs, q)
        if ftype == 4:
            f.set_bs_coeff(f0, fs, q)
        assert np.abs(f.m_coeff - scipy.signal.firwin(taps, fs=fs, **kw)).max() < 1e-15
        assert np.abs(f.m_coeff - oracle.fir_design(taps, ftype, f0, fs, q)).max() < 1e-15
    h = np.zeros(64)
    assert lib.sdsp_hip_fir_design(64, 2, 10e3, 100e3, 0.0, 1.0, h.ctypes.data) == L.ERR_INVALID_SIZE  # even taps, passes fs/2
    assert lib.sdsp_hip_fir_design(0, 1, 10e3, 100e3, 0.0, 1.0, h.ctypes.data) == L.ERR_INVALID_SIZE
    assert lib.sdsp_hip_fir_design(31, 1, 60e3, 100e3, 0.0, 1.0, h.ctypes.data) == L.ERR_INVALID_ARG   # beyond fs/2
    assert lib.sdsp_hip_fir_design(31, 0, 10e3, 100e3, 0.0, 1.0, h.ctypes.data) == L.ERR_INVALID_ARG
    plan = C.c_void_p()
    assert lib.sdsp_hip_fir_plan_create(C.byref(plan), 0, h.ctypes.data, L.F32, 0) == L.ERR_INVALID_SIZE
    assert lib.sdsp_hip_fir_plan_create(C.byref(plan), 5000, h.ctypes.data, L.F32, 0) == L.ERR_INVALID_SIZE


def test_design_matches_reference_fixtures(iir_golden):
    for tag in iir_golden["csv_names"]:
        ftype, fs, f0, q = iir_golden[f"{tag}__params"]
        f = sd.casc_2o_iir(4)
        design(f, int(ftype), f0, fs, q)
        assert np.array_equal(f.m_a_coeff, iir_golden[f"{tag}__a"])
        assert np.array_equal(f.m_b_coeff, iir_golden[f"{tag}__b"])
        assert f.m_gain == float(iir_golden[f"{tag}__gain"])


def test_preload_matches_reference_fixtures(iir_golden):
    lib = sd.load()
    for nm, ftype in (("lp", 1), ("hp", 2), ("bp", 3)):
        f = sd.casc_2o_iir(4)
        design(f, ftype, 10e3, 100e3, 1.1)
        mem = np.zeros((5, 3))
        L.check(lib.sdsp_hip_iir_preload(4, f.m_f_type, f.m_a_coeff.ctypes.data, f.m_b_coeff.ctypes.data,
                                         f.m_gain, 10.0, mem.ctypes.data))
        assert np.array_equal(mem, iir_golden[f"preload_{nm}__mem"])


def test_error_codes_replace_static_asserts():
    lib = sd.load()
    h = C.c_void_p()
    # radix 0 = auto still validates the size before any device is touched
    assert lib.sdsp_hip_fft_plan_create(C.byref(h), 96, 0, 1, 0, 1, 0) == L.ERR_INVALID_SIZE
    # invalid sizes are rejected before any device is touched: fft.h:261,304
    assert lib.sdsp_hip_fft_plan_create(C.byref(h), 96, 2, 1, 0, 1, 0) == L.ERR_INVALID_SIZE
    assert b"power of 2" in lib.sdsp_hip_last_error_string()
    assert lib.sdsp_hip_fft_plan_create(C.byref(h), 2048, 4, 1, 0, 1, 0) == L.ERR_INVALID_SIZE
    assert b"power of 4" in lib.sdsp_hip_last_error_string()
    assert lib.sdsp_hip_fft_plan_create(C.byref(h), 64, 3, 1, 0, 1, 0) == L.ERR_UNSUPPORTED
    assert lib.sdsp_hip_fft_plan_create(C.byref(h), 64, 2, 0, 0, 1, 0) == L.ERR_INVALID_ARG
    a = np.zeros(9)
    g = C.c_double()
    assert lib.sdsp_hip_iir_design_lp(3, 1e3, 1e4, 1.0, a.ctypes.data, a.ctypes.data, C.byref(g)) == L.ERR_INVALID_SIZE
    assert b"M must be even" in lib.sdsp_hip_last_error_string()  # casc_2o_iir.h:25
    with pytest.raises(ValueError):
        sd.casc_2o_iir(3)


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the fail-loudly path is exercised on CPU-only hosts")
    with pytest.raises(sd.SdspHipError) as e:
        sd.FftPlan(4096, 4)
    assert e.value.code == L.ERR_NO_DEVICE
    with pytest.raises(sd.SdspHipError):
        sd.fft_radix2(np.zeros(64, np.complex128))
    n = C.c_int(-1)
    assert sd.load().sdsp_hip_device_count(C.byref(n)) == 0 and n.value == 0
    plan = C.c_void_p()
    h = np.ones(8)
    assert sd.load().sdsp_hip_fir_plan_create(C.byref(plan), 8, h.ctypes.data, L.F64, 0) == L.ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under simpledsp_amd/ or include/ may reference it
    for p in list((ROOT / "simpledsp_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if p.is_file() and p.suffix in {".py", ".h", ".hip", ".cpp", ".hpp"}:
            text = p.read_text()
            assert "sdsp_oracle" not in text and "import oracle" not in text and "from oracle" not in text, p


def test_a_touched_kernel_header_makes_its_objects_stale():
    """build hygiene (round 2 verdict, weak #7): dependencies come from the compiler's own -MMD lists, so editing
    csrc/fft32_r4.h -- which only fft_big.hip includes -- rebuilds fft_big.o and nothing that does not include it"""
    import os
    from simpledsp_amd import build as B
    B.build_library()
    hdr = B.CSRC / "fft32_r4.h"
    big, iir = B.OBJ_DIR / "fft_big.o", B.OBJ_DIR / "iir.o"
    assert hdr in (B._dep_files(big.with_suffix(".d")) or [])
    assert not B.object_stale(big, B.CSRC / "fft_big.hip")
    st = hdr.stat()
    try:
        os.utime(hdr, (st.st_atime, big.stat().st_mtime + 10))
        assert B.object_stale(big, B.CSRC / "fft_big.hip")
        assert not B.object_stale(iir, B.CSRC / "iir.hip")
    finally:
        os.utime(hdr, (st.st_atime, st.st_mtime))
    assert not B.object_stale(big, B.CSRC / "fft_big.hip")


def test_loader_refuses_a_library_built_from_other_sources(monkeypatch):
    """the in-tree .so is what travels to the GPU box: one whose embedded source hash differs from the tree is refused"""
    import simpledsp_amd._lib as L
    from simpledsp_amd import build as B
    lib = L.load()
    assert L.built_hash(lib) == B.source_hash()
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(B, "source_hash", lambda: "0123456789abcdef")
    monkeypatch.setenv("SDSP_HIP_NO_REBUILD", "1")
    with pytest.raises(L.StaleLibraryError):
        L.load()


def test_source_hash_through_a_symlinked_checkout(tmp_path):
    """the GPU box reaches the tree through a symlink (/root/repo -> scratch copy): the hash must not depend on the path
    the package was imported by (a relative_to() on an unresolved __file__ raised there once)"""
    import subprocess
    import sys
    link = tmp_path / "repo_link"
    link.symlink_to(ROOT, target_is_directory=True)
    code = ("import sys; sys.path.insert(0, %r); from simpledsp_amd import build as B; print(B.source_hash())" % str(link))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    from simpledsp_amd import build as B
    assert r.stdout.strip() == B.source_hash()


@pytest.mark.parametrize("src,flags", [("fft_big64.hip", []), ("fft_2pass.hip", [])])
def test_no_wide_store_is_followed_by_a_write_to_its_data_registers(src, flags):
    """profiles/r03_store_hazard.md: a vector-memory store of more than 64 bits whose data register is overwritten by the NEXT
    instruction needs a wait state; the compiler inserts it for global stores and for buffer stores with an immediate soffset, not
    for buffer stores whose soffset is an SGPR -- which is how fft_big64.hip once produced doubles with an LDS address as their low
    word on some boxes.  tools/isa_store_hazard.py scans the device assembly for the pattern (no GPU needed)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "isa_store_hazard.py"), str(ROOT / "simpledsp_amd" / "csrc" / src), *flags],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "unguarded overwrites of store data: 0" in r.stdout
