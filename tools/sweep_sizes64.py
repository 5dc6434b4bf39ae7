#!/usr/bin/env python3
"""f64 throughput per size: register-pass f64 kernels (variant 0) vs the coverage kernel (variant 1)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

total = 1 << 26  # complex128 elements = 1 GiB
buf = torch.view_as_complex(torch.randn((total, 2), device="cuda", dtype=torch.float64))
for n in (64, 1024, 4096, 8192):
    batch = total // n
    x = buf.view(batch, n)
    for variant, name in ((0, "reg f64"), (1, "coverage")):
        fwd = sd.FftPlan(n, 2, sd.forward_fft, sd.F64, max_batch=64); rev = sd.FftPlan(n, 2, sd.reverse_fft, sd.F64, max_batch=64)
        fwd.set_variant(variant); rev.set_variant(variant)
        for _ in range(2):
            fwd.exec(x); rev.exec(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fwd.exec(x); rev.exec(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        print(f"N={n:5d} f64 radix 2 {name:9s}: {ms:7.3f} ms per GiB -> {batch/ms/1e3:8.2f} M FFT/s, {2*total*16/ms/1e6:6.0f} GB/s ({2*total*16/ms/1e6/80:.1f} %)")
