#!/usr/bin/env python3
"""tools/conv_probe.py [n=4096] [batch=262144] [radix] [variant=0] -- fused fast convolution (sdsp_hip_fft_convolve) on a large
batch, one launch against the library's launch pieces (sdsp_hip_set_launch_piece_bytes).  variant 2: the register-pass
family's fused kernel where a one-wave kernel is the default; 1: three launches."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import simpledsp_amd as sd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
sd.load()
x = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda", dtype=torch.float32))
h = torch.view_as_complex(torch.randn((n, 2), device="cuda", dtype=torch.float32) * (1.0 / n ** 0.5))
radix = int(sys.argv[3]) if len(sys.argv) > 3 else (4 if sd.isPowerOf4(n) else 2)
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
plan = sd.FftPlan(n, radix, sd.forward_fft, sd.F32, max_batch=batch)
plan.set_variant(variant)
default = sd.get_launch_piece_bytes()
for piece in (0, default, 0, default):
    sd.set_launch_piece_bytes(piece)
    for _ in range(5):
        plan.convolve(x, h)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.convolve(x, h)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"N = {n} radix {radix} variant {variant}, {batch} rows, piece {piece >> 20} MiB: {ms:.3f} ms, {batch / ms / 1e3:.1f} M convolutions/s, "
          f"{batch * n * 16 / (ms * 1e-3) / 8e12 * 100:.1f} % of HBM peak on its compulsory bytes", flush=True)
