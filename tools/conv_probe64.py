#!/usr/bin/env python3
"""Fused convolution in double: fft_big64.hip's form (variant 0) against the register-pass family's MODE 3 (variant 2, N <= 8192) and three
launches (variant 1); 1 GiB batches, % of HBM peak on the compulsory bytes (read + write once)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd
for rep in range(2):
    for n in (4096, 8192, 16384):
        batch = (1 << 26) // n
        x = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda", dtype=torch.float64))
        ph = torch.rand((n,), device="cuda", dtype=torch.float64) * 6.283185307179586
        h = torch.polar(torch.ones_like(ph), ph)
        for variant in (0, 2, 1):
            if variant == 2 and n > 8192:
                continue
            p = sd.FftPlan(n, 2, sd.forward_fft, sd.F64, max_batch=batch); p.set_variant(variant)
            for _ in range(2):
                p.convolve(x, h)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                p.convolve(x, h)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 6
            print(f"round {rep} N={n} f64 conv variant {variant}: {ms:.3f} ms per GiB, {2*(1<<30)/ms/1e6/80:.1f} % of HBM peak, {batch/ms/1e3:.2f} M conv/s", flush=True)
            del p
