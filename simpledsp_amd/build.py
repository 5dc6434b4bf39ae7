"""Build libsdsp_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Objects go to build/obj (git-ignored); the shared library is written in-tree to
simpledsp_amd/lib/ so that it travels with the source snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
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
    "fft_big64.hip": ["-fno-slp-vectorize"],
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


def source_hash() -> str:
    """sha256 over everything the library is built from: csrc/*, include/**, and this file's flags"""
    h = hashlib.sha256()
    # (everything through the resolved package path: the tree may be reached through a symlink, as /root/repo is on the GPU box)
    files = sorted(CSRC.glob("*")) + sorted((ROOT / "include").rglob("*.h")) + [Path(__file__).resolve()]
    for f in files:
        if f.is_file():
            h.update(f.relative_to(ROOT).as_posix().encode())
            h.update(b"\0")
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _dep_files(dfile: Path):
    """prerequisites listed in a compiler-written Makefile fragment (-MMD -MF), or None when it is missing"""
    if not dfile.exists():
        return None
    text = dfile.read_text().replace("\\\n", " ")
    deps = []
    for line in text.splitlines():
        if ":" in line:
            deps += line.split(":", 1)[1].split()
    return [Path(d) for d in deps]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any((not Path(d).exists()) or Path(d).stat().st_mtime > t for d in deps)


def object_stale(obj: Path, src: Path) -> bool:
    """an object is rebuilt when the compiler's own dependency list for it (every header it included, transitively) has a
    newer file, or when that list does not exist yet; build.py itself (the flags) counts as a prerequisite"""
    deps = _dep_files(obj.with_suffix(".d"))
    if deps is None:
        return True
    return _stale(obj, [src, Path(__file__).resolve(), *deps])


def build_library(force: bool = False, verbose: bool = False) -> Path:
    cc = hipcc()
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    digest = source_hash()
    stamp = OBJ_DIR / "source_hash.txt"
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        obj = OBJ_DIR / (src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        flags = list(extra)
        stale = force or object_stale(obj, CSRC / src)
        if src == "host_math.cpp":  # carries the hash: rebuilt whenever any source changed
            flags.append(f'-DSDSP_HIP_SOURCE_HASH="{digest}"')
            stale = stale or not stamp.exists() or stamp.read_text().strip() != digest
        if stale:
            # -MMD -MF: hipcc compiles host and device in one go; the host pass writes the dependency file
            jobs.append([cc, *COMMON, *flags, "-MMD", "-MF", str(obj.with_suffix(".d")), "-c", str(CSRC / src), "-o", str(obj)])

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
    stamp.write_text(digest + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
