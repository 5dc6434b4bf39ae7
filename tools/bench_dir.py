#!/usr/bin/env python3
"""Steady-state throughput of one plan direction: tools/bench_dir.py N RADIX [GiB]  (forward, reverse separately)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

n, radix = int(sys.argv[1]), int(sys.argv[2])
gib = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
total = int(gib * (1 << 27))
buf = torch.view_as_complex(torch.randn((total, 2), device="cuda"))
x = buf.view(total // n, n)
for name, T in (("forward", sd.forward_fft), ("reverse", sd.reverse_fft), ("forward", sd.forward_fft), ("reverse", sd.reverse_fft)):
    plan = sd.FftPlan(n, radix, T, sd.F32, max_batch=16)
    buf.normal_()
    for _ in range(30):
        plan.exec(x)
    torch.cuda.synchronize()
    buf.normal_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(12):  # few enough that repeated unscaled forward transforms stay finite in f32
        plan.exec(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 12
    print(f"N={n} radix {radix} {name} [{plan.info.kernel.decode()}] {gib:g} GiB: {ms:.4f} ms, {2*total*8/ms/1e6/80:.1f} % of 8 TB/s", flush=True)
