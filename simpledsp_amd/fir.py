"""Host mirror of the FIR filter bank (include/sdsp_hip.h).

The reference has no FIR filter -- it is a TODO in its README (README.md:16) -- so this class follows
the conventions of its IIR class (casc_2o_iir.h:23-37: set_*_coeff(f0, fs[, q], gain_in), process in
place, preload_filter, copy_coeff_from) rather than mirroring existing code."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class fir_filter:
    """A bank of `channels` identical n_taps-tap direct-form FIR filters with per-channel history."""

    def __init__(self, n_taps: int, channels: int = 1, precision: int = L.F32, device: int = 0):
        if n_taps <= 0:
            raise ValueError("n_taps must be positive")
        self._lib = L.load()
        self.n_taps, self.channels, self.precision, self.device = n_taps, channels, precision, device
        self.m_coeff = np.zeros(n_taps)
        self.m_f_type = L.FILTER_NONE
        self._plan = None
        self._state = None  # torch tensor (channels, n_taps-1), newest input first
        self._variant = 0

    # ---- design (host, double): Hamming-windowed sinc == scipy.signal.firwin
    def _design(self, f_type, f0, fs, q, gain_in):
        L.check(self._lib.sdsp_hip_fir_design(self.n_taps, f_type, f0, fs, q, gain_in, self.m_coeff.ctypes.data))
        self.m_f_type = f_type
        self._drop_plan()

    def set_lp_coeff(self, f0, fs, gain_in=1.0):
        self._design(L.FILTER_LOW_PASS, f0, fs, 0.0, gain_in)

    def set_hp_coeff(self, f0, fs, gain_in=1.0):
        self._design(L.FILTER_HIGH_PASS, f0, fs, 0.0, gain_in)

    def set_bp_coeff(self, f0, fs, q, gain_in=1.0):
        self._design(L.FILTER_BAND_PASS, f0, fs, q, gain_in)

    def set_bs_coeff(self, f0, fs, q, gain_in=1.0):
        self._design(L.FILTER_BAND_STOP, f0, fs, q, gain_in)

    def set_coeff(self, h):
        h = np.asarray(h, dtype=np.float64).reshape(-1)
        if h.size != self.n_taps:
            raise ValueError("coefficient count differs from n_taps")
        self.m_coeff = h.copy()
        self.m_f_type = L.FILTER_NONE
        self._drop_plan()

    def copy_coeff_from(self, other: "fir_filter"):  # design, not history (casc_2o_iir.h:28-34 semantics)
        self.set_coeff(other.m_coeff)
        self.m_f_type = other.m_f_type

    def preload_filter(self, value: float):  # history of a steady input (casc_2o_iir.h:197-214 semantics)
        import torch
        dt = torch.float64 if self.precision == L.F64 else torch.float32
        self._state = torch.full((self.channels, max(self.n_taps - 1, 1)), value, dtype=dt, device=f"cuda:{self.device}")

    def set_variant(self, v: int):
        self._variant = v
        if self._plan:
            L.check(self._lib.sdsp_hip_fir_plan_set_variant(self._plan, v))

    def reset(self):
        self._state = None

    @property
    def state(self):
        return self._state

    def _drop_plan(self):
        if self._plan:
            self._lib.sdsp_hip_fir_plan_destroy(self._plan)
            self._plan = None

    def _ensure_plan(self):
        if self._plan is None:
            h = C.c_void_p()
            L.check(self._lib.sdsp_hip_fir_plan_create(C.byref(h), self.n_taps, self.m_coeff.ctypes.data, self.precision,
                                                       self.device))
            self._plan = h
            L.check(self._lib.sdsp_hip_fir_plan_set_variant(h, self._variant))

    def process(self, data, samples: int | None = None, offset: int = 0):
        """data: contiguous device tensor (channels, stride); filters data[:, offset:offset+samples] of every
        channel in place, continuing from the bank's history."""
        import torch
        dt = torch.float64 if self.precision == L.F64 else torch.float32
        if data.dtype != dt or not data.is_cuda or not data.is_contiguous() or data.dim() != 2:
            raise ValueError("process needs a contiguous (channels, samples) device tensor of the bank dtype")
        if data.shape[0] != self.channels:
            raise ValueError("channel count differs from the bank's")
        if data.device.index != self.device:
            raise ValueError("tensor lives on a different device than the bank")
        stride = data.shape[1]
        samples = stride - offset if samples is None else samples
        if offset + samples > stride:
            raise ValueError("block exceeds the row")
        self._ensure_plan()
        if self._state is None:
            self._state = torch.zeros((self.channels, max(self.n_taps - 1, 1)), dtype=dt, device=f"cuda:{self.device}")
        stream = torch.cuda.current_stream(data.device).cuda_stream
        L.check(self._lib.sdsp_hip_fir_process(self._plan, data.data_ptr() + offset * data.element_size(), self.channels,
                                               samples, stride, self._state.data_ptr(), stream))
        return data

    def __del__(self):
        try:
            self._drop_plan()
        except Exception:
            pass
