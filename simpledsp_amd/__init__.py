"""simpledsp_amd -- MI355X-native batched FFT + cascaded-biquad engine behind simpledsp's sdsp:: surface.

The product is the HIP library (csrc/ -> lib/libsdsp_hip.so) behind the C ABI in include/sdsp_hip.h;
this package is its Python host mirror.  Importing does not touch the GPU; any compute call without
the library or without a HIP device raises (there is no CPU path here).
"""
from ._lib import (F32, F64, F32_F64STATE, FORWARD, REVERSE, IIR_GENERIC, IIR_LP, IIR_HP, IIR_BP, FILTER_NONE,
                   FILTER_LOW_PASS, FILTER_HIGH_PASS, FILTER_BAND_PASS, FILTER_BAND_STOP, SdspHipError, load)
from .fft import (FftPlan, RfftPlan, fft_radix2, fft_radix4, forward_fft, reverse_fft, log2, log4, isPowerOf2,
                  isPowerOf4, digit_reverse, calc_swap_lookup, calc_twiddles, calc_wCoeffs)
from .iir import casc_2o_iir, casc_2o_iir_lp, casc_2o_iir_hp, casc_2o_iir_bp
from .fir import fir_filter


def set_launch_piece_bytes(nbytes: int) -> None:
    """Process-wide launch granularity (include/sdsp_hip.h: sdsp_hip_set_launch_piece_bytes); 0 = never split."""
    from ._lib import check
    check(load().sdsp_hip_set_launch_piece_bytes(int(nbytes)))


def get_launch_piece_bytes() -> int:
    import ctypes
    from ._lib import check
    v = ctypes.c_uint64(0)
    check(load().sdsp_hip_get_launch_piece_bytes(ctypes.byref(v)))
    return v.value


class filter_type:  # filter_type.h:6
    none, low_pass, high_pass, band_pass = 0, 1, 2, 3
    band_stop = 4  # not in the reference's enum (README.md:15 TODO)
