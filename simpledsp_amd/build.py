"""Build libsdsp_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Objects go to build/obj (git-ignored); the shared library is written in-tree to
simpledsp_amd/lib/ so that it travels with the source snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB_PATH = LIB_DIR / "libsdsp_hip.so"
OBJ_DIR = ROOT / "build" / "obj"

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", f"-I{ROOT / 'include'}", f"-I{CSRC}",
          "-Wall", "-Wno-unused-function"]
# per-source extra flags.  -fno-slp-vectorize: packing scalar f32 math into v_pk_*_f32 buys nothing on
# CDNA4's SIMD-32 and costs v_mov shuffles + register-pair constraints (measured: fft4096 +1.8 %,
# fft_reg up to +7 %, fft1m +17 % and no scratch, iir -800 v_mov per kernel).  iir.hip keeps the reference's operation order (no FMA contraction) so
# that the f64 kernel reproduces casc_2o_iir.h bit for bit.
SOURCES = {
    "host_math.cpp": ["-x", "hip"],
    "capi.hip": [],
    "fft_tile.hip": [],
    "fft4096.hip": ["-fno-slp-vectorize"],
    "fft1m.hip": ["-fno-slp-vectorize"],  # SLP packing cost 44-76 B/lane of scratch here
    "fft_reg.hip": ["-fno-slp-vectorize"],
    "fft_reg64.hip": ["-fno-slp-vectorize"],
    "fft_big.hip": ["-fno-slp-vectorize"],
    "fft_mix.hip": ["-fno-slp-vectorize"],
    "fft_wave.hip": ["-fno-slp-vectorize"],
    "fft_mid.hip": ["-fno-slp-vectorize"],
    "fft_2pass.hip": ["-fno-slp-vectorize"],
    "iir.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "fir.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],  # f64 taps: multiply then add, like the oracle
}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libsdsp_hip.so)")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> Path:
    cc = hipcc()
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    headers = [CSRC / "sdsp_hip_internal.h", CSRC / "fft_passes.h", CSRC / "fft32.h", CSRC / "fft1m_kernels.h", CSRC / "fft4096_kernels.h", ROOT / "include" / "sdsp_hip.h", Path(__file__)]
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        obj = OBJ_DIR / (src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [CSRC / src, *headers]):
            jobs.append([cc, *COMMON, *extra, "-c", str(CSRC / src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB_PATH), "-lpthread"])
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
