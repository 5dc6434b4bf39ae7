"""The index-arithmetic models of the registers-resident FFT kernels (csrc/fft_big.hip, csrc/fft32_r4.h) run on the CPU:
the cheap LDS address forms against the swizzle they replace, and the radix-4 layered dataflow (layers, rotations, every
multiplier formed as table value x compile-time constant) against numpy.fft -- host logic, no GPU."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("script", ["model_fft_big_lds.py", "model_fft_big_r4.py 14", "model_fft_big_r4.py 14 rev", "model_fft_big_r4.py 12",
                                    "model_fft_big_r4.py 12 rev"])
def test_model_script_passes(script):
    name, *rest = script.split()
    r = subprocess.run([sys.executable, str(ROOT / "tools" / name), *rest], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "equal sw<" in r.stdout or "max rel err" in r.stdout
