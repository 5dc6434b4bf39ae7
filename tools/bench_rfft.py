#!/usr/bin/env python3
"""Real-input packing vs the complex transform of the same real signals, 1 GiB of reals.
   tools/bench_rfft.py [n_real,n_real,...] [radix of the real plans]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

radix = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for n_real in ([int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else (1024, 8192)):
    batch = (1 << 28) // n_real  # 1 GiB of float32
    x = torch.randn((batch, n_real), device="cuda")
    fwd, inv = sd.RfftPlan(n_real, radix, sd.forward_fft, batch), sd.RfftPlan(n_real, radix, sd.reverse_fft, batch)
    xc = torch.view_as_complex(torch.randn((batch, n_real, 2), device="cuda"))
    cf, cr = sd.FftPlan(n_real, 2, sd.forward_fft, sd.F32, batch), sd.FftPlan(n_real, 2, sd.reverse_fft, sd.F32, batch)
    for name, run in (("real-packed", lambda: (fwd.exec(x), inv.exec(x))), ("complex", lambda: (cf.exec(xc), cr.exec(xc)))):
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        byts = batch * n_real * (4 if name == "real-packed" else 8) * 2
        print(f"n_real={n_real:6d} {name:12s}: {ms:.3f} ms per {batch} transforms -> {batch/ms/1e3:8.1f} M transforms/s, {byts/ms/1e6:6.0f} GB/s")
