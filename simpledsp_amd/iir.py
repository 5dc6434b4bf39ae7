"""Host mirror of sdsp::casc_2o_iir<m_t> and casc_2o_iir_{lp,hp,bp}<m_t> (casc_2o_iir.h) for banks
of channels on the MI355X.  Same method names and argument meaning as the reference; `process`
filters `channels` independent streams in place through the C ABI (include/sdsp_hip.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class casc_2o_iir:
    """A bank of `channels` identical m_t-section cascades with per-channel state.

    kind = IIR_GENERIC is casc_2o_iir<m_t>; IIR_LP/HP/BP are the numerator-folded classes.
    State lives on the device between process() calls (block streaming is bit-identical to one
    long call, testIIR.cpp:61-75)."""

    def __init__(self, m_t: int = 4, channels: int = 1, precision: int = L.F32, kind: int = L.IIR_GENERIC,
                 device: int = 0):
        if m_t % 2 != 0 or m_t <= 0:
            raise ValueError("M must be even!")  # static_assert casc_2o_iir.h:25
        self._lib = L.load()
        self.m_t, self.channels, self.precision, self.kind, self.device = m_t, channels, precision, kind, device
        self.m_gain = 1.0
        self.m_a_coeff = np.zeros((m_t, 3))
        self.m_b_coeff = np.zeros((m_t, 3))
        self.m_f_type = L.FILTER_NONE
        self._plan = None
        self._state = None  # torch tensor (3*(m_t+1), channels) on the device, or None = zeros
        self._variant = 0

    def _dtypes(self):
        """(sample dtype, state dtype): F32_F64STATE keeps float samples and a double recurrence"""
        import torch
        return (torch.float64 if self.precision == L.F64 else torch.float32,
                torch.float32 if self.precision == L.F32 else torch.float64)

    # ---- coefficient design (host, double): casc_2o_iir.h:82-194
    def _designed(self, f_type):
        self.m_f_type = f_type
        self._drop_plan()

    def set_lp_coeff(self, f0, fs, gain_in=1.0):
        g = C.c_double()
        L.check(self._lib.sdsp_hip_iir_design_lp(self.m_t, f0, fs, gain_in, self.m_a_coeff.ctypes.data,
                                                 self.m_b_coeff.ctypes.data, C.byref(g)))
        self.m_gain = g.value
        self._designed(L.FILTER_LOW_PASS)

    def set_hp_coeff(self, f0, fs, gain_in=1.0):
        g = C.c_double()
        L.check(self._lib.sdsp_hip_iir_design_hp(self.m_t, f0, fs, gain_in, self.m_a_coeff.ctypes.data,
                                                 self.m_b_coeff.ctypes.data, C.byref(g)))
        self.m_gain = g.value
        self._designed(L.FILTER_HIGH_PASS)

    def set_bp_coeff(self, f0, fs, q, gain_in=1.0):
        g = C.c_double()
        L.check(self._lib.sdsp_hip_iir_design_bp(self.m_t, f0, fs, q, gain_in, self.m_a_coeff.ctypes.data,
                                                 self.m_b_coeff.ctypes.data, C.byref(g)))
        self.m_gain = g.value
        self._designed(L.FILTER_BAND_PASS)

    def set_bs_coeff(self, f0, fs, q, gain_in=1.0):  # README.md:15 TODO in the reference; generic kind only
        g = C.c_double()
        L.check(self._lib.sdsp_hip_iir_design_bs(self.m_t, f0, fs, q, gain_in, self.m_a_coeff.ctypes.data,
                                                 self.m_b_coeff.ctypes.data, C.byref(g)))
        self.m_gain = g.value
        self._designed(L.FILTER_BAND_STOP)

    def copy_coeff_from(self, other: "casc_2o_iir"):  # casc_2o_iir.h:28-34: design, not state
        self.m_gain = other.m_gain
        self.m_a_coeff = other.m_a_coeff.copy()
        self.m_b_coeff = other.m_b_coeff.copy()
        self.m_f_type = other.m_f_type
        self._drop_plan()

    def preload_filter(self, value: float):  # casc_2o_iir.h:197-214, every channel
        import torch
        mem = np.zeros((self.m_t + 1, 3))
        L.check(self._lib.sdsp_hip_iir_preload(self.m_t, self.m_f_type, self.m_a_coeff.ctypes.data,
                                               self.m_b_coeff.ctypes.data, self.m_gain, value, mem.ctypes.data))
        dt = self._dtypes()[1]
        col = torch.from_numpy(mem.reshape(-1).copy()).to(dt)
        self._state = col[:, None].expand(-1, self.channels).contiguous().to(f"cuda:{self.device}")

    def set_variant(self, v: int):
        self._variant = v
        if self._plan:
            L.check(self._lib.sdsp_hip_iir_plan_set_variant(self._plan, v))

    # ---- state
    def _drop_plan(self):
        if self._plan:
            self._lib.sdsp_hip_iir_plan_destroy(self._plan)
            self._plan = None

    def _ensure_plan(self):
        if self._plan is None:
            h = C.c_void_p()
            L.check(self._lib.sdsp_hip_iir_plan_create(C.byref(h), self.m_t, self.kind, self.m_a_coeff.ctypes.data,
                                                       self.m_b_coeff.ctypes.data, self.m_gain, self.precision,
                                                       self.device))
            self._plan = h
            L.check(self._lib.sdsp_hip_iir_plan_set_variant(h, self._variant))

    def reset(self):
        self._state = None

    def kernel_name(self, data, samples: int | None = None, offset: int = 0) -> str:
        """the kernel process(data, samples, offset) would launch (the library's own selection function)"""
        self._ensure_plan()
        stride = data.shape[1]
        samples = stride - offset if samples is None else samples
        buf = C.create_string_buffer(64)
        L.check(self._lib.sdsp_hip_iir_plan_kernel(self._plan, data.data_ptr() + offset * data.element_size(), self.channels,
                                                   samples, stride, buf, 64))
        return buf.value.decode()

    @property
    def state(self):
        """(3*(m_t+1), channels): row 3*j+age = level j's value `age+1` samples ago."""
        return self._state

    # ---- process(): casc_2o_iir.h:36-80 / :228-263
    def process(self, data, samples: int | None = None, offset: int = 0):
        """data: contiguous device tensor (channels, stride); filters data[:, offset:offset+samples]
        of every channel in place, continuing from the bank's state."""
        import torch
        dt, st = self._dtypes()
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
            self._state = torch.zeros((3 * (self.m_t + 1), self.channels), dtype=st, device=f"cuda:{self.device}")
        stream = torch.cuda.current_stream(data.device).cuda_stream
        L.check(self._lib.sdsp_hip_iir_process(self._plan, data.data_ptr() + offset * data.element_size(),
                                               self.channels, samples, stride, self._state.data_ptr(), stream))
        return data

    def process_interleaved(self, data, samples: int | None = None, offset: int = 0):
        """data: contiguous device tensor (samples, channels) -- the sample-major "wire" layout
        (SURVEY 8f-2).  Filters rows offset..offset+samples in place, continuing from the bank's
        state; bit-identical to process() on the transposed data, without any transpose."""
        import torch
        dt, st = self._dtypes()
        if data.dtype != dt or not data.is_cuda or not data.is_contiguous() or data.dim() != 2:
            raise ValueError("process_interleaved needs a contiguous (samples, channels) device tensor")
        if data.shape[1] != self.channels:
            raise ValueError("channel count differs from the bank's")
        if data.device.index != self.device:
            raise ValueError("tensor lives on a different device than the bank")
        samples = data.shape[0] - offset if samples is None else samples
        if offset + samples > data.shape[0]:
            raise ValueError("block exceeds the buffer")
        self._ensure_plan()
        if self._state is None:
            self._state = torch.zeros((3 * (self.m_t + 1), self.channels), dtype=st, device=f"cuda:{self.device}")
        stream = torch.cuda.current_stream(data.device).cuda_stream
        L.check(self._lib.sdsp_hip_iir_process_interleaved(
            self._plan, data.data_ptr() + offset * self.channels * data.element_size(), self.channels, samples,
            self.channels, self._state.data_ptr(), stream))
        return data

    def __del__(self):
        try:
            self._drop_plan()
        except Exception:
            pass


def casc_2o_iir_lp(m_t=4, channels=1, precision=L.F32, device=0):
    return casc_2o_iir(m_t, channels, precision, L.IIR_LP, device)


def casc_2o_iir_hp(m_t=4, channels=1, precision=L.F32, device=0):
    return casc_2o_iir(m_t, channels, precision, L.IIR_HP, device)


def casc_2o_iir_bp(m_t=4, channels=1, precision=L.F32, device=0):
    return casc_2o_iir(m_t, channels, precision, L.IIR_BP, device)
