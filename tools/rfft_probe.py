#!/usr/bin/env python3
"""tools/rfft_probe.py [n_real=2048] [radix=2] -- real-input packing (sdsp_hip_rfft_plan_*) on 4 GiB of reals: the size's
default kernel (variant 0) against the register-pass family's in-LDS split (variant 1)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import simpledsp_amd as sd

n_real = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
radix = int(sys.argv[2]) if len(sys.argv) > 2 else 2
batch = (1 << 30) // n_real
sd.load()
x = torch.randn((batch, n_real), device="cuda", dtype=torch.float32)
fwd = sd.RfftPlan(n_real, radix, sd.forward_fft, max_batch=batch)
inv = sd.RfftPlan(n_real, radix, sd.reverse_fft, max_batch=batch)
for variant in (0, 1, 2, 0, 1, 2):
    fwd.set_variant(variant)
    inv.set_variant(variant)
    for _ in range(5):
        fwd.exec(x)
        inv.exec(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fwd.exec(x)
        inv.exec(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    print(f"n_real = {n_real} radix {radix} variant {variant} ({fwd.info.kernel.decode()}): {ms:.3f} ms per pass over 4 GiB, "
          f"{batch / ms / 1e3:.1f} M transforms/s, {batch * n_real * 8 / (ms * 1e-3) / 8e12 * 100:.1f} % of HBM peak", flush=True)
