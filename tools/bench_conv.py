#!/usr/bin/env python3
"""Fast convolution throughput, N=4096 f32, batch 65536: fused kernel vs three launches."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

import sys as _sys
n = int(_sys.argv[1]) if len(_sys.argv) > 1 else 4096
radix = int(_sys.argv[2]) if len(_sys.argv) > 2 else 4
batch = (1 << 28) // n
x = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda"))
h = torch.view_as_complex(torch.randn((n, 2), device="cuda"))
h = h / h.abs()  # unit-modulus response keeps repeated convolution bounded
plan = sd.FftPlan(n, radix, sd.forward_fft, sd.F32, max_batch=batch)
print(f"N={n} radix {radix} batch {batch}")
for variant, name in ((0, "fused single kernel"), (1, "forward + multiply + reverse")):
    plan.set_variant(variant)
    for _ in range(30):
        plan.convolve(x, h)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        plan.convolve(x, h)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{name:32s}: {ms:.3f} ms per {batch} convolutions -> {batch/ms/1e3:.1f} M conv/s; "
          f"compulsory bytes at {batch*n*16/ms/1e6:.0f} GB/s = {batch*n*16/ms/1e6/80:.1f} % of 8 TB/s")
