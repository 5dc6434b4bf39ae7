"""oracle -- CPU checker for the simpledsp FFT / cascaded-biquad hot path.

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never from the product package ``simpledsp_amd``.

Two back ends, both double precision on the host CPU:

* ``Oracle``  -- ``_build/libsdsp_oracle.so``, the plain-C restatement (oracle/sdsp_oracle.c),
  available everywhere (the GPU box included).
* ``Reference`` -- ``_ref/libsdsp_ref.so``, the REAL reference headers behind a C shim
  (oracle/ref_shim.cpp).  Buildable only where ``/root/reference`` exists; the built library
  travels to the GPU box, the sources do not.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
ORACLE_LIB = _HERE / "_build" / "libsdsp_oracle.so"
REF_LIB = _HERE / "_ref" / "libsdsp_ref.so"
MAX_SECTIONS = 32

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint)


def build(ref: bool = True) -> None:
    """Compile the checker libraries (building the checker is not using it)."""
    targets = ["_build/libsdsp_oracle.so"] + (["ref"] if ref else [])
    subprocess.run(["make", "-C", str(_HERE), *targets], check=True,
                   stdout=subprocess.DEVNULL)


class _IirStruct(C.Structure):
    _fields_ = [
        ("m", C.c_uint),
        ("pos", C.c_int),
        ("gain", C.c_double),
        ("f_type", C.c_int),
        ("mem", C.c_double * ((MAX_SECTIONS + 1) * 3)),
        ("b", C.c_double * (MAX_SECTIONS * 3)),
        ("a", C.c_double * (MAX_SECTIONS * 3)),
    ]


class Oracle:
    """ctypes face of oracle/sdsp_oracle.c."""

    def __init__(self):
        if not ORACLE_LIB.exists():
            build(ref=False)
        self.lib = lib = C.CDLL(str(ORACLE_LIB))
        lib.sdsp_oracle_fft_plan_create.restype = C.c_void_p
        lib.sdsp_oracle_fft_plan_create.argtypes = [C.c_uint, C.c_int, C.c_int]
        lib.sdsp_oracle_fft_plan_destroy.argtypes = [C.c_void_p]
        lib.sdsp_oracle_fft_exec.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.sdsp_oracle_calc_wcoeffs.argtypes = [C.c_uint, C.c_int, C.c_void_p]
        lib.sdsp_oracle_calc_swap_lookup.argtypes = [C.c_uint, C.c_uint, C.c_void_p]
        lib.sdsp_oracle_digit_reverse.restype = C.c_uint
        lib.sdsp_oracle_digit_reverse.argtypes = [C.c_uint, C.c_uint, C.c_uint]
        for name in ("log2", "log4"):
            f = getattr(lib, f"sdsp_oracle_{name}")
            f.restype = C.c_uint
            f.argtypes = [C.c_uint]
        S = C.POINTER(_IirStruct)
        lib.sdsp_oracle_iir_init.argtypes = [S, C.c_uint]
        lib.sdsp_oracle_iir_copy_coeff_from.argtypes = [S, S]
        lib.sdsp_oracle_iir_set_lp_coeff.argtypes = [S, C.c_double, C.c_double, C.c_double]
        lib.sdsp_oracle_iir_set_hp_coeff.argtypes = [S, C.c_double, C.c_double, C.c_double]
        lib.sdsp_oracle_iir_set_bp_coeff.argtypes = [S, C.c_double, C.c_double, C.c_double, C.c_double]
        lib.sdsp_oracle_iir_set_bs_coeff.argtypes = [S, C.c_double, C.c_double, C.c_double, C.c_double]
        lib.sdsp_oracle_iir_preload_filter.argtypes = [S, C.c_double]
        lib.sdsp_oracle_fir_design.argtypes = [C.c_uint, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
        lib.sdsp_oracle_fir_process.argtypes = [C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.sdsp_oracle_fir_process.restype = None
        lib.sdsp_oracle_iir_process.argtypes = [S, C.c_int, C.c_void_p, C.c_size_t]
        self._plans: dict = {}

    # ---- helpers -------------------------------------------------------------------
    def log2(self, v): return self.lib.sdsp_oracle_log2(v)
    def log4(self, v): return self.lib.sdsp_oracle_log4(v)
    def is_power_of_2(self, v): return bool(self.lib.sdsp_oracle_is_power_of_2(v))
    def is_power_of_4(self, v): return bool(self.lib.sdsp_oracle_is_power_of_4(v))
    def digit_reverse(self, n, base, x): return self.lib.sdsp_oracle_digit_reverse(n, base, x)

    def wcoeffs(self, n: int, reverse: bool = False) -> np.ndarray:
        out = np.empty((self.log2(n), n), dtype=np.complex128)
        rc = self.lib.sdsp_oracle_calc_wcoeffs(n, int(reverse), out.ctypes.data)
        if rc:
            raise ValueError(f"calc_wcoeffs({n}) -> {rc}")
        return out

    def swap_lookup(self, n: int, base: int) -> np.ndarray:
        out = np.empty(n, dtype=np.uint32)
        rc = self.lib.sdsp_oracle_calc_swap_lookup(n, base, out.ctypes.data)
        if rc:
            raise ValueError(f"calc_swap_lookup({n},{base}) -> {rc}")
        return out

    # ---- FFT -----------------------------------------------------------------------
    def _plan(self, n, radix, reverse):
        key = (n, radix, bool(reverse))
        if key not in self._plans:
            p = self.lib.sdsp_oracle_fft_plan_create(n, radix, int(reverse))
            if not p:
                raise ValueError(f"fft size {n} is not valid for radix {radix}")
            self._plans[key] = p
        return self._plans[key]

    def fft(self, x, radix: int, reverse: bool = False) -> np.ndarray:
        """x: (..., n) complex; returns the reference algorithm's result in double."""
        a = np.array(x, dtype=np.complex128, order="C", copy=True)
        n = a.shape[-1]
        batch = a.size // n if n else 0
        rc = self.lib.sdsp_oracle_fft_exec(self._plan(n, radix, reverse), a.ctypes.data, batch)
        if rc:
            raise RuntimeError(f"fft_exec -> {rc}")
        return a

    def fft_inplace(self, a: np.ndarray, radix: int, reverse: bool = False) -> None:
        """timing entry: a is C-contiguous complex128 (batch, n), transformed in place."""
        assert a.dtype == np.complex128 and a.flags.c_contiguous
        n = a.shape[-1]
        self.lib.sdsp_oracle_fft_exec(self._plan(n, radix, reverse), a.ctypes.data, a.size // n)

    # ---- IIR -----------------------------------------------------------------------
    def iir(self, m: int = 4) -> "OracleIir":
        return OracleIir(self, m)

    # ---- FIR (README.md:16 TODO in the reference: no reference code, pinned to scipy) -----------
    def fir_design(self, taps: int, filter_type: int, f0: float, fs: float, q: float = 0.0, gain_in: float = 1.0):
        h = np.zeros(taps)
        if self.lib.sdsp_oracle_fir_design(taps, filter_type, f0, fs, q, gain_in, h.ctypes.data):
            raise ValueError("fir_design: bad arguments")
        return h

    def fir_process(self, h, data, mem=None):
        """returns (filtered copy, final history); mem = previous inputs, newest first (taps-1 values)"""
        h = np.ascontiguousarray(h, dtype=np.float64)
        a = np.array(data, dtype=np.float64, order="C", copy=True)
        m = np.zeros(max(h.size - 1, 1)) if mem is None else np.array(mem, dtype=np.float64, copy=True)
        self.lib.sdsp_oracle_fir_process(h.size, h.ctypes.data, m.ctypes.data, a.ctypes.data, a.size)
        return a, m[: h.size - 1]

    def __del__(self):
        try:
            for p in self._plans.values():
                self.lib.sdsp_oracle_fft_plan_destroy(p)
        except Exception:
            pass


class OracleIir:
    """sdsp::casc_2o_iir<m> (kind 0) / casc_2o_iir_{lp,hp,bp}<m> (kind 1/2/3) restated."""

    def __init__(self, oracle: Oracle, m: int):
        self._o = oracle
        self._s = _IirStruct()
        if oracle.lib.sdsp_oracle_iir_init(C.byref(self._s), m):
            raise ValueError("M must be even!")
        self.m = m

    def copy(self) -> "OracleIir":
        o = OracleIir(self._o, self.m)
        C.memmove(C.byref(o._s), C.byref(self._s), C.sizeof(_IirStruct))
        return o

    def copy_coeff_from(self, other: "OracleIir"):
        self._o.lib.sdsp_oracle_iir_copy_coeff_from(C.byref(self._s), C.byref(other._s))

    def set_lp_coeff(self, f0, fs, gain_in=1.0):
        self._o.lib.sdsp_oracle_iir_set_lp_coeff(C.byref(self._s), f0, fs, gain_in)

    def set_hp_coeff(self, f0, fs, gain_in=1.0):
        self._o.lib.sdsp_oracle_iir_set_hp_coeff(C.byref(self._s), f0, fs, gain_in)

    def set_bp_coeff(self, f0, fs, q, gain_in=1.0):
        self._o.lib.sdsp_oracle_iir_set_bp_coeff(C.byref(self._s), f0, fs, q, gain_in)

    def set_bs_coeff(self, f0, fs, q, gain_in=1.0):
        # not in the reference (README TODO); pinned to scipy in tests/test_oracle_iir.py
        self._o.lib.sdsp_oracle_iir_set_bs_coeff(C.byref(self._s), f0, fs, q, gain_in)

    def preload_filter(self, value):
        self._o.lib.sdsp_oracle_iir_preload_filter(C.byref(self._s), value)

    def process(self, data, kind: int = 0) -> np.ndarray:
        a = np.array(data, dtype=np.float64, order="C", copy=True)
        rc = self._o.lib.sdsp_oracle_iir_process(C.byref(self._s), kind, a.ctypes.data, a.size)
        if rc:
            raise ValueError(f"iir_process -> {rc}")
        return a

    def set_design(self, a, b, gain, f_type=0):
        """load given coefficients (the recurrence of casc_2o_iir.h:36-80 does not care where they came from)"""
        a = np.asarray(a, dtype=np.float64).reshape(-1)
        b = np.asarray(b, dtype=np.float64).reshape(-1)
        assert a.size == b.size == 3 * self.m
        for i in range(3 * self.m):
            self._s.a[i] = a[i]
            self._s.b[i] = b[i]
        self._s.gain = float(gain)
        self._s.f_type = int(f_type)

    def process_inplace(self, a: np.ndarray, kind: int = 0) -> None:
        assert a.dtype == np.float64 and a.flags.c_contiguous
        self._o.lib.sdsp_oracle_iir_process(C.byref(self._s), kind, a.ctypes.data, a.size)

    @property
    def gain(self): return self._s.gain
    @property
    def f_type(self): return self._s.f_type
    @property
    def pos(self): return self._s.pos
    @property
    def a(self): return np.array(self._s.a[: self.m * 3]).reshape(self.m, 3)
    @property
    def b(self): return np.array(self._s.b[: self.m * 3]).reshape(self.m, 3)
    @property
    def mem(self): return np.array(self._s.mem[: (self.m + 1) * 3]).reshape(self.m + 1, 3)


class Reference:
    """ctypes face of oracle/ref_shim.cpp (the real simpledsp headers). Sizes 4..4096."""

    SIZES = (4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096)

    def __init__(self):
        if not REF_LIB.exists():
            raise FileNotFoundError(
                f"{REF_LIB} not built (needs /root/reference; run `make -C oracle ref`)")
        self.lib = lib = C.CDLL(str(REF_LIB))
        lib.sdsp_ref_fft.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        lib.sdsp_ref_wcoeffs.argtypes = [C.c_uint, C.c_int, C.c_void_p]
        lib.sdsp_ref_swap_lookup.argtypes = [C.c_uint, C.c_uint, C.c_void_p]
        for name in ("log2", "log4"):
            f = getattr(lib, f"sdsp_ref_{name}")
            f.restype = C.c_uint
            f.argtypes = [C.c_uint]
        lib.sdsp_ref_iir_create.restype = C.c_void_p
        lib.sdsp_ref_iir_create.argtypes = [C.c_uint, C.c_int]
        lib.sdsp_ref_iir_clone.restype = C.c_void_p
        lib.sdsp_ref_iir_clone.argtypes = [C.c_void_p]
        lib.sdsp_ref_iir_destroy.argtypes = [C.c_void_p]
        lib.sdsp_ref_iir_set_lp_coeff.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        lib.sdsp_ref_iir_set_hp_coeff.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        lib.sdsp_ref_iir_set_bp_coeff.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        lib.sdsp_ref_iir_preload_filter.argtypes = [C.c_void_p, C.c_double]
        lib.sdsp_ref_iir_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.sdsp_ref_iir_coeffs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.sdsp_ref_iir_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]

    @staticmethod
    def available() -> bool:
        return REF_LIB.exists()

    def log2(self, v): return self.lib.sdsp_ref_log2(v)
    def log4(self, v): return self.lib.sdsp_ref_log4(v)
    def is_power_of_2(self, v): return bool(self.lib.sdsp_ref_is_power_of_2(v))
    def is_power_of_4(self, v): return bool(self.lib.sdsp_ref_is_power_of_4(v))

    def wcoeffs(self, n, reverse=False):
        out = np.empty((self.log2(n), n), dtype=np.complex128)
        rc = self.lib.sdsp_ref_wcoeffs(n, int(reverse), out.ctypes.data)
        if rc:
            raise ValueError(f"ref wcoeffs({n}) -> {rc}")
        return out

    def swap_lookup(self, n, base):
        out = np.empty(n, dtype=np.uint32)
        rc = self.lib.sdsp_ref_swap_lookup(n, base, out.ctypes.data)
        if rc:
            raise ValueError(f"ref swap_lookup({n},{base}) -> {rc}")
        return out

    def fft(self, x, radix, reverse=False):
        a = np.array(x, dtype=np.complex128, order="C", copy=True)
        n = a.shape[-1]
        rc = self.lib.sdsp_ref_fft(n, radix, int(reverse), a.ctypes.data, a.size // n)
        if rc:
            raise ValueError(f"ref fft(n={n}, radix={radix}) -> {rc}")
        return a

    def fft_inplace(self, a, radix, reverse=False):
        assert a.dtype == np.complex128 and a.flags.c_contiguous
        n = a.shape[-1]
        self.lib.sdsp_ref_fft(n, radix, int(reverse), a.ctypes.data, a.size // n)

    def iir(self, m=4, kind=0):
        return ReferenceIir(self, m, kind)


class ReferenceIir:
    def __init__(self, ref: Reference, m: int, kind: int, _h=None):
        self._r = ref
        self.m, self.kind = m, kind
        self._h = _h if _h is not None else ref.lib.sdsp_ref_iir_create(m, kind)
        if not self._h:
            raise ValueError(f"reference shim has no casc_2o_iir<{m}> kind {kind}")

    def copy(self):
        return ReferenceIir(self._r, self.m, self.kind, self._r.lib.sdsp_ref_iir_clone(self._h))

    def _chk(self, rc, what):
        if rc:
            raise ValueError(f"{what} not available on kind {self.kind}")

    def set_lp_coeff(self, f0, fs, gain_in=1.0):
        self._chk(self._r.lib.sdsp_ref_iir_set_lp_coeff(self._h, f0, fs, gain_in), "set_lp_coeff")

    def set_hp_coeff(self, f0, fs, gain_in=1.0):
        self._chk(self._r.lib.sdsp_ref_iir_set_hp_coeff(self._h, f0, fs, gain_in), "set_hp_coeff")

    def set_bp_coeff(self, f0, fs, q, gain_in=1.0):
        self._chk(self._r.lib.sdsp_ref_iir_set_bp_coeff(self._h, f0, fs, q, gain_in), "set_bp_coeff")

    def preload_filter(self, v):
        self._chk(self._r.lib.sdsp_ref_iir_preload_filter(self._h, v), "preload_filter")

    def process(self, data, kind=None):
        a = np.array(data, dtype=np.float64, order="C", copy=True)
        self._r.lib.sdsp_ref_iir_process(self._h, a.ctypes.data, a.size)
        return a

    def process_inplace(self, a, kind=None):
        assert a.dtype == np.float64 and a.flags.c_contiguous
        self._r.lib.sdsp_ref_iir_process(self._h, a.ctypes.data, a.size)

    def coeffs(self):
        a = np.empty((self.m, 3))
        b = np.empty((self.m, 3))
        g = C.c_double()
        self._chk(self._r.lib.sdsp_ref_iir_coeffs(self._h, a.ctypes.data, b.ctypes.data, C.byref(g)), "coeffs")
        return a, b, g.value

    def state(self):
        mem = np.empty((self.m + 1, 3))
        pos = C.c_int()
        self._chk(self._r.lib.sdsp_ref_iir_state(self._h, mem.ctypes.data, C.byref(pos)), "state")
        return mem, pos.value

    def __del__(self):
        try:
            self._r.lib.sdsp_ref_iir_destroy(self._h)
        except Exception:
            pass
