"""Host mirror of the reference's FFT surface (include/sdsp/fft.h of simpledsp) for Python.

Same names and argument meaning as the reference where it has them (fft_radix2 / fft_radix4,
forward_fft / reverse_fft, log2 / log4 / isPowerOf2 / isPowerOf4, digit_reverse, calc_wCoeffs),
batched and running on the MI355X through the C ABI (include/sdsp_hip.h).  torch is used for
device memory and streams only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class forward_fft:  # fft.h:135-146
    direction = L.FORWARD

    @staticmethod
    def Sign() -> float:
        return 1.0


class reverse_fft:  # fft.h:121-133
    direction = L.REVERSE

    @staticmethod
    def Sign() -> float:
        return -1.0


def log2(num: int) -> int:  # fft.h:12-19
    return L.load().sdsp_hip_log2(num)


def log4(num: int) -> int:  # fft.h:21-28
    return L.load().sdsp_hip_log4(num)


def isPowerOf2(num: int) -> bool:  # fft.h:31-37
    return bool(L.load().sdsp_hip_is_power_of_2(num))


def isPowerOf4(num: int) -> bool:  # fft.h:40-43
    return bool(L.load().sdsp_hip_is_power_of_4(num))


def digit_reverse(n: int, base: int, x: int) -> int:  # fft.h:217-236
    return L.load().sdsp_hip_digit_reverse(n, base, x)


def calc_swap_lookup(n: int, base: int) -> np.ndarray:  # fft.h:238-256
    lut = np.array([digit_reverse(n, base, i) for i in range(n)], dtype=np.uint32)
    for i in range(1, n - 1):
        i2 = lut[i]
        if i2 != i:
            lut[i2] = i2
    return lut


def calc_twiddles(n: int, T=forward_fft) -> np.ndarray:
    """Row exp(-/+2*pi*i*j/n), j<n: the run-time form of calc_wCoeffs' last row (fft.h:197-214)."""
    out = np.empty(n, dtype=np.complex128)
    L.check(L.load().sdsp_hip_calc_twiddles(n, T.direction, out.ctypes.data))
    return out


def calc_wCoeffs(n: int, T=forward_fft) -> np.ndarray:
    """coeff_array<N> (fft.h:48-49): row i, column j = exp(-/+2*pi*i*j / 2^(i+1))."""
    w = calc_twiddles(n, T)
    rows = log2(n)
    j = np.arange(n)
    return np.stack([w[(j * (n >> (i + 1))) % n] for i in range(rows)])


_TORCH_DTYPES = None


def _torch_dtypes():
    global _TORCH_DTYPES
    if _TORCH_DTYPES is None:
        import torch
        _TORCH_DTYPES = {torch.complex64: L.F32, torch.complex128: L.F64}
    return _TORCH_DTYPES


class FftPlan:
    """A batched transform plan: twiddles precomputed in double, resident in HBM."""

    def __init__(self, n: int, radix: int, T=forward_fft, precision: int = L.F32, max_batch: int = 1,
                 device: int = 0):
        self._lib = L.load()
        self._h = C.c_void_p()
        L.check(self._lib.sdsp_hip_fft_plan_create(C.byref(self._h), n, radix, T.direction, precision,
                                                    max_batch, device))
        self.n, self.radix, self.direction, self.precision, self.device = n, radix, T.direction, precision, device
        if radix == 0:  # SDSP_HIP_RADIX_AUTO: the library chose
            self.radix = int(self.info.radix)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdsp_hip_fft_plan_destroy(self._h)
            self._h = None

    __del__ = close

    def set_variant(self, v: int):
        L.check(self._lib.sdsp_hip_fft_plan_set_variant(self._h, v))

    def status(self):
        """synchronise and raise if the plan's last launch gave up on a bounded in-kernel wait (N = 2^20 persistent kernel)"""
        L.check(self._lib.sdsp_hip_fft_plan_status(self._h))

    def set_wait_limit(self, ticks: int):
        """testing hook: bound of the N = 2^20 kernel's hand-off waits in 100 MHz ticks (default 2 s)"""
        L.check(self._lib.sdsp_hip_fft_plan_set_wait_limit(self._h, ticks))

    def launches(self, batch: int) -> int:
        """kernel launches one exec of `batch` transforms issues (launch pieces and workspace slices included)"""
        n = C.c_uint64()
        L.check(self._lib.sdsp_hip_fft_plan_launches(self._h, batch, C.byref(n)))
        return int(n.value)

    @property
    def info(self) -> L.PlanInfo:
        info = L.PlanInfo()
        L.check(self._lib.sdsp_hip_fft_plan_get_info(self._h, C.byref(info)))
        return info

    def twiddles(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.complex128 if self.precision == L.F64 else np.complex64)
        L.check(self._lib.sdsp_hip_fft_plan_get_twiddles(self._h, out.ctypes.data))
        return out

    def exec_ptr(self, data_ptr: int, batch: int, stream: int = 0):
        L.check(self._lib.sdsp_hip_fft_exec(self._h, data_ptr, batch, stream))

    def exec(self, x):
        """In place on a contiguous CUDA/HIP torch tensor (..., n), on torch's current stream."""
        import torch
        prec = _torch_dtypes().get(x.dtype)
        if prec != self.precision or not x.is_cuda or not x.is_contiguous() or x.shape[-1] != self.n:
            raise ValueError("exec needs a contiguous device tensor (..., n) of the plan's complex dtype")
        if x.device.index != self.device:
            raise ValueError("tensor lives on a different device than the plan")
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self.exec_ptr(x.data_ptr(), x.numel() // self.n, stream)
        return x

    def convolve(self, x, h):
        """Fast convolution: x <- IFFT(FFT(x) * h) per transform, in place.  x: contiguous
        device tensor (..., n); h: device tensor (n,) -- the frequency response.  Needs a forward plan."""
        import torch
        if _torch_dtypes().get(x.dtype) != self.precision or h.dtype != x.dtype or not x.is_cuda or not h.is_cuda:
            raise ValueError("convolve needs device tensors of the plan's complex dtype")
        if not x.is_contiguous() or not h.is_contiguous() or x.shape[-1] != self.n or h.numel() != self.n:
            raise ValueError("convolve needs contiguous x (..., n) and h (n,)")
        if x.device.index != self.device or h.device.index != self.device:
            raise ValueError("tensors live on a different device than the plan")
        stream = torch.cuda.current_stream(x.device).cuda_stream
        L.check(self._lib.sdsp_hip_fft_convolve(self._h, x.data_ptr(), h.data_ptr(), x.numel() // self.n, stream))
        return x

    def exec_host(self, a: np.ndarray) -> np.ndarray:
        """In place on a C-contiguous numpy array (..., n): H2D, transform, D2H."""
        want = np.complex128 if self.precision == L.F64 else np.complex64
        if a.dtype != want or not a.flags.c_contiguous or a.shape[-1] != self.n:
            raise ValueError("exec_host needs a C-contiguous (..., n) array of the plan's complex dtype")
        L.check(self._lib.sdsp_hip_fft_exec_host(self._h, a.ctypes.data, a.size // self.n))
        return a


class RfftPlan:
    """Real-input packing: n_real real samples <-> packed half spectrum, in place.

    forward_fft: float32 (..., n_real) -> the same memory viewed as complex64 (..., n_real/2) with
    out[..., k] = X[k] (0 < k < n_real/2) and out[..., 0] = X[0] + 1j*X[n_real/2]; reverse_fft inverts."""

    def __init__(self, n_real: int, radix: int = 2, T=forward_fft, max_batch: int = 1, device: int = 0, precision: int = L.F32):
        self._lib = L.load()
        self._h = C.c_void_p()
        L.check(self._lib.sdsp_hip_rfft_plan_create_p(C.byref(self._h), n_real, radix, T.direction, precision, max_batch, device))
        self.n_real, self.radix, self.direction, self.device, self.precision = n_real, radix, T.direction, device, precision

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdsp_hip_fft_plan_destroy(self._h)
            self._h = None

    __del__ = close

    def set_variant(self, v: int):  # 0: the size's default kernel; 1 / 2: the register-pass family (A/B, cross-checks)
        L.check(self._lib.sdsp_hip_fft_plan_set_variant(self._h, v))

    def launches(self, batch: int) -> int:
        n = C.c_uint64()
        L.check(self._lib.sdsp_hip_fft_plan_launches(self._h, batch, C.byref(n)))
        return int(n.value)

    @property
    def info(self) -> L.PlanInfo:
        i = L.PlanInfo()
        L.check(self._lib.sdsp_hip_fft_plan_get_info(self._h, C.byref(i)))
        return i

    def exec(self, x):
        """x: contiguous float32 device tensor (..., n_real), transformed in place; returns the complex
        view (forward) or x itself (reverse: pass the float32 view of the packed spectrum)."""
        import torch
        want = torch.float64 if self.precision == L.F64 else torch.float32
        if x.dtype != want or not x.is_cuda or not x.is_contiguous() or x.shape[-1] != self.n_real:
            raise ValueError("exec needs a contiguous device tensor (..., n_real) of the plan's real dtype")
        if x.device.index != self.device:
            raise ValueError("tensor lives on a different device than the plan")
        stream = torch.cuda.current_stream(x.device).cuda_stream
        L.check(self._lib.sdsp_hip_fft_exec(self._h, x.data_ptr(), x.numel() // self.n_real, stream))
        if self.direction == L.FORWARD:
            return torch.view_as_complex(x.view(*x.shape[:-1], self.n_real // 2, 2))
        return x


_plan_cache: dict = {}


def _cached_plan(n, radix, T, precision, device) -> FftPlan:
    key = (n, radix, T.direction, precision, device)
    if key not in _plan_cache:
        _plan_cache[key] = FftPlan(n, radix, T, precision, max_batch=1, device=device)
    return _plan_cache[key]


def _dispatch(radix, data, T):
    if isinstance(data, np.ndarray):
        prec = {np.dtype(np.complex64): L.F32, np.dtype(np.complex128): L.F64}.get(data.dtype)
        if prec is None:
            raise ValueError("data must be complex64 or complex128")
        return _cached_plan(data.shape[-1], radix, T, prec, 0).exec_host(data)
    prec = _torch_dtypes().get(data.dtype)
    if prec is None:
        raise ValueError("data must be complex64 or complex128")
    return _cached_plan(data.shape[-1], radix, T, prec, data.device.index or 0).exec(data)


def fft_radix2(data, T=forward_fft):
    """sdsp::fft_radix2<T,N>(data) (fft.h:258-299) over a batch (..., N), in place."""
    return _dispatch(2, data, T)


def fft_radix4(data, T=forward_fft):
    """sdsp::fft_radix4<T,N>(data) (fft.h:301-360) over a batch (..., N), in place."""
    return _dispatch(4, data, T)
