"""The C++ drop-in surface (include/sdsp/*.h): the reference's own tests restated in
tests/cpp/test_dropin.cpp, compiled with g++ and the reference's warning flags against this
repository's headers and libsdsp_hip.so."""
import subprocess

import pytest

from conftest import GOLDEN, ROOT

BIN = ROOT / "build" / "test_dropin"


def _build():
    import simpledsp_amd
    simpledsp_amd.load()  # builds libsdsp_hip.so if needed
    r = subprocess.run(["make", "-C", str(ROOT / "tests" / "cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "warning" not in r.stderr, r.stderr  # -Wall -Wextra -Wpedantic -Wconversion clean
    return BIN


def test_headers_compile_and_fail_loudly_without_gpu():
    import torch
    exe = _build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_dropin_program_on_gpu")
    r = subprocess.run([str(exe), str(GOLDEN / "impulse_response")], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr
    assert "no CPU fallback" in r.stdout


def test_headers_also_compile_with_clang():
    # the reference's fft.h needs GCC (constexpr std::sin/cos, its README.md:20); this one does not
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra",
                        f"-I{ROOT / 'include'}", str(ROOT / "tests" / "cpp" / "test_dropin.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_dropin_program_on_gpu():
    exe = _build()
    r = subprocess.run([str(exe), str(GOLDEN / "impulse_response")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout
