"""AddressSanitizer + UBSan over the CPU restatement (the reference's only safety net is .at()
bounds checks; GPU sanitizers are not available on the pool)."""
import subprocess

from conftest import ROOT


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "asan_check"
    cmd = ["gcc", "-O1", "-g", "-std=gnu11", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", str(ROOT / "oracle" / "asan_check.c"), str(ROOT / "oracle" / "sdsp_oracle.c"),
           "-I", str(ROOT / "oracle"), "-o", str(exe), "-lquadmath", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sanitizers clean" in r.stdout
