"""ctypes binding of libsdsp_hip.so -- the C ABI declared in include/sdsp_hip.h.

The library is the product: if it cannot be loaded this module raises.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "lib" / "libsdsp_hip.so"

F32, F64 = 0, 1
F32_F64STATE = 2  # IIR banks: float samples, double state / recurrence
FORWARD, REVERSE = 1, -1
FILTER_NONE, FILTER_LOW_PASS, FILTER_HIGH_PASS, FILTER_BAND_PASS = 0, 1, 2, 3
FILTER_BAND_STOP = 4
IIR_GENERIC, IIR_LP, IIR_HP, IIR_BP = 0, 1, 2, 3
MAX_SECTIONS = 16

OK, ERR_INVALID_SIZE, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE, ERR_INVALID_ARG, ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6


class SdspHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"sdsp_hip error {code}: {message}")
        self.code = code
        self.message = message


class PlanInfo(C.Structure):
    _fields_ = [
        ("n", C.c_uint32), ("radix", C.c_int), ("direction", C.c_int), ("precision", C.c_int),
        ("device", C.c_int), ("hbm_passes", C.c_int), ("algorithmic_bytes", C.c_uint64),
        ("workspace_bytes", C.c_uint64), ("twiddle_bytes", C.c_uint64), ("kernel", C.c_char * 64),
        ("stage_radix", C.c_int),
    ]


# name -> (restype, argtypes); every symbol include/sdsp_hip.h declares
_vp, _u32, _u64, _i, _d, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double, C.c_size_t
_pp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "sdsp_hip_last_error_string": (C.c_char_p, []),
    "sdsp_hip_version": (C.c_char_p, []),
    "sdsp_hip_device_count": (_i, [C.POINTER(_i)]),
    "sdsp_hip_malloc": (_i, [_pp, _sz, _i]),
    "sdsp_hip_free": (_i, [_vp, _i]),
    "sdsp_hip_memcpy_h2d": (_i, [_vp, _vp, _sz, _i]),
    "sdsp_hip_memcpy_d2h": (_i, [_vp, _vp, _sz, _i]),
    "sdsp_hip_device_synchronize": (_i, [_i]),
    "sdsp_hip_log2": (C.c_uint, [C.c_uint]),
    "sdsp_hip_log4": (C.c_uint, [C.c_uint]),
    "sdsp_hip_is_power_of_2": (_i, [C.c_uint]),
    "sdsp_hip_is_power_of_4": (_i, [C.c_uint]),
    "sdsp_hip_digit_reverse": (C.c_uint, [C.c_uint, C.c_uint, C.c_uint]),
    "sdsp_hip_calc_twiddles": (_i, [C.c_uint, _i, _vp]),
    "sdsp_hip_fft_plan_create": (_i, [_pp, _u32, _i, _i, _i, _u64, _i]),
    "sdsp_hip_rfft_plan_create": (_i, [_pp, _u32, _i, _i, _u64, _i]),
    "sdsp_hip_rfft_plan_create_p": (_i, [_pp, _u32, _i, _i, _i, _u64, _i]),
    "sdsp_hip_fft_plan_destroy": (_i, [_vp]),
    "sdsp_hip_fft_exec": (_i, [_vp, _vp, _u64, _vp]),
    "sdsp_hip_fft_exec_host": (_i, [_vp, _vp, _u64]),
    "sdsp_hip_fft_exec_sharded": (_i, [_pp, _i, _vp, _u64]),
    "sdsp_hip_fft_convolve": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "sdsp_hip_fft_plan_get_info": (_i, [_vp, C.POINTER(PlanInfo)]),
    "sdsp_hip_fft_plan_get_twiddles": (_i, [_vp, _vp]),
    "sdsp_hip_fft_plan_set_variant": (_i, [_vp, _i]),
    "sdsp_hip_fft_plan_status": (_i, [_vp]),
    "sdsp_hip_fft_plan_set_wait_limit": (_i, [_vp, _u64]),
    "sdsp_hip_fft_plan_launches": (_i, [_vp, _u64, C.POINTER(_u64)]),
    "sdsp_hip_set_launch_piece_bytes": (_i, [_u64]),
    "sdsp_hip_get_launch_piece_bytes": (_i, [C.POINTER(_u64)]),
    "sdsp_hip_iir_design_lp": (_i, [_u32, _d, _d, _d, _vp, _vp, C.POINTER(_d)]),
    "sdsp_hip_iir_design_hp": (_i, [_u32, _d, _d, _d, _vp, _vp, C.POINTER(_d)]),
    "sdsp_hip_iir_design_bp": (_i, [_u32, _d, _d, _d, _d, _vp, _vp, C.POINTER(_d)]),
    "sdsp_hip_iir_design_bs": (_i, [_u32, _d, _d, _d, _d, _vp, _vp, C.POINTER(_d)]),
    "sdsp_hip_iir_preload": (_i, [_u32, _i, _vp, _vp, _d, _d, _vp]),
    "sdsp_hip_iir_plan_create": (_i, [_pp, _u32, _i, _vp, _vp, _d, _i, _i]),
    "sdsp_hip_iir_plan_destroy": (_i, [_vp]),
    "sdsp_hip_iir_process": (_i, [_vp, _vp, _u64, _u64, _u64, _vp, _vp]),
    "sdsp_hip_iir_process_interleaved": (_i, [_vp, _vp, _u64, _u64, _u64, _vp, _vp]),
    "sdsp_hip_iir_process_host": (_i, [_vp, _vp, _u64, _u64, _u64, _vp]),
    "sdsp_hip_iir_process_sharded": (_i, [_pp, _i, _vp, _u64, _u64]),
    "sdsp_hip_iir_state_bytes": (_i, [_vp, _u64, C.POINTER(_u64)]),
    "sdsp_hip_iir_plan_set_variant": (_i, [_vp, _i]),
    "sdsp_hip_iir_plan_kernel": (_i, [_vp, _vp, _u64, _u64, _u64, C.c_char_p, _sz]),
    "sdsp_hip_fir_design": (_i, [_u32, _i, _d, _d, _d, _d, _vp]),
    "sdsp_hip_fir_plan_create": (_i, [_pp, _u32, _vp, _i, _i]),
    "sdsp_hip_fir_plan_destroy": (_i, [_vp]),
    "sdsp_hip_fir_process": (_i, [_vp, _vp, _u64, _u64, _u64, _vp, _vp]),
    "sdsp_hip_fir_process_host": (_i, [_vp, _vp, _u64, _u64, _u64, _vp]),
    "sdsp_hip_fir_state_bytes": (_i, [_vp, _u64, C.POINTER(_u64)]),
    "sdsp_hip_fir_plan_set_variant": (_i, [_vp, _i]),
}

_lib = None


class StaleLibraryError(RuntimeError):
    """libsdsp_hip.so does not match the sources beside it"""


def built_hash(lib) -> str:
    """the source hash the library was built from (sdsp_hip_version() ends in "src:<hash>")"""
    lib.sdsp_hip_version.restype = C.c_char_p
    v = lib.sdsp_hip_version().decode()
    return v.rsplit("src:", 1)[1] if "src:" in v else "unhashed"


def _open(path, fresh: bool = False) -> C.CDLL:
    if fresh:  # dlopen caches by path: load the rebuilt file through a private copy
        import shutil
        import tempfile
        tmp = Path(tempfile.mkdtemp(prefix="sdsp_hip_")) / path.name
        shutil.copy2(path, tmp)
        path = tmp
    return C.CDLL(str(path))


def _bind(lib) -> C.CDLL:
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load libsdsp_hip.so (building it with hipcc first if it is not there)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  A process must end up
    # with ONE HIP runtime: if torch is going to be used (it is our device-memory plumbing), it has
    # to be loaded first so that this library binds to the same copy by soname.  Loading ours first
    # leaves two HSA runtimes in the process and the second one finds no device.
    if os.environ.get("SDSP_HIP_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    from .build import build_library, source_hash
    if not LIB_PATH.exists():
        if not build_if_missing:
            raise FileNotFoundError(f"{LIB_PATH} is missing: run `python -m simpledsp_amd.build`")
        build_library()
    lib = _open(LIB_PATH)
    # a library built from other sources than the ones beside it (an edited header whose object was not rebuilt, a .so
    # left over from another checkout) must not be what gets tested and benched: rebuild, or refuse
    want = source_hash()
    if built_hash(lib) != want:
        if not build_if_missing or os.environ.get("SDSP_HIP_NO_REBUILD") == "1":
            raise StaleLibraryError(f"{LIB_PATH} was built from other sources (library {built_hash(lib)}, tree {want}): "
                                    "run `python -m simpledsp_amd.build`")
        print(f"simpledsp_amd: {LIB_PATH.name} is stale ({built_hash(lib)} != {want}), rebuilding", file=sys.stderr)
        build_library()
        lib = _open(LIB_PATH, fresh=True)
        if built_hash(lib) != want:
            raise StaleLibraryError(f"{LIB_PATH} still reports {built_hash(lib)} after a rebuild (tree {want})")
    _lib = _bind(lib)
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise SdspHipError(rc, load().sdsp_hip_last_error_string().decode(errors="replace"))
