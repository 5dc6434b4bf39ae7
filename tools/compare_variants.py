#!/usr/bin/env python3
"""Steady-state comparison of kernel variants of one plan: tools/compare_variants.py N RADIX V0,V1,... [GiB]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

n, radix = int(sys.argv[1]), int(sys.argv[2])
variants = [int(v) for v in sys.argv[3].split(",")]
gib = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
total = int(gib * (1 << 27))
buf = torch.view_as_complex(torch.randn((total, 2), device="cuda"))
x = buf.view(total // n, n)
plan = sd.FftPlan(n, radix, sd.forward_fft, sd.F32, max_batch=16)
back = sd.FftPlan(n, radix, sd.reverse_fft, sd.F32, max_batch=16)
# forward / reverse alternate so the in-place data stays finite: repeated unscaled forward transforms
# overflow to NaN after a few dozen launches, and kernels measured up to 5 % FASTER on all-NaN data
# (lower switching power) -- an artefact this tool must not report
for rep in range(2):
    for v in variants:
        plan.set_variant(v)
        back.set_variant(v)
        for _ in range(15):
            plan.exec(x); back.exec(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.exec(x); back.exec(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        print(f"N={n} radix {radix} variant {v} [{plan.info.kernel.decode()}]: {ms:.4f} ms per {gib:g} GiB, {2*total*8/ms/1e6/80:.1f} % of 8 TB/s", flush=True)
        buf.normal_()
