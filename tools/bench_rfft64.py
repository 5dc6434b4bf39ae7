#!/usr/bin/env python3
"""Real-input plans in double: fft_big64.hip's REAL form (variant 0) against the register-pass family's MODE 1 / 2 (variant 1, n_real <= 16384) and
the complex path on the same samples; 1 GiB of real samples, forward / inverse alternating; % of HBM peak on the compulsory bytes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd
for rep in range(2):
    for n_real in (8192, 16384, 32768):
        batch = (1 << 27) // n_real
        x = torch.randn((batch, n_real), device="cuda", dtype=torch.float64)
        for variant in (0, 1):
            if variant == 1 and n_real > 16384:
                continue
            f = sd.RfftPlan(n_real, 2, sd.forward_fft, max_batch=batch, precision=sd.F64); i = sd.RfftPlan(n_real, 2, sd.reverse_fft, max_batch=batch, precision=sd.F64)
            f.set_variant(variant); i.set_variant(variant)
            for _ in range(2):
                s = f.exec(x); i.exec(torch.view_as_real(s).reshape(batch, n_real))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                s = f.exec(x); i.exec(torch.view_as_real(s).reshape(batch, n_real))
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 8
            print(f"round {rep} n_real={n_real} f64 variant {variant} [{f.info.kernel.decode()}]: {ms:.3f} ms per GiB, {2*(1<<30)/ms/1e6/80:.1f} % of HBM peak, {batch/ms/1e3:.2f} M transforms/s", flush=True)
