#!/usr/bin/env python3
"""Rank kernel variants on the GPU: interleaved rounds in ONE process, median of HIP-event times
(cdna_hip_programming.md rule 24).  Usage: python tools/sweep.py fft4096|iir [--variants 0,1,2]"""
import argparse
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["fft4096", "iir", "iir64"])
ap.add_argument("--variants", default="")
ap.add_argument("--rounds", type=int, default=15)
ap.add_argument("--units", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")

if args.what == "fft4096":
    batch = args.units or 65536
    x = torch.view_as_complex(torch.randn((batch, 4096, 2), device=dev))
    fwd = sd.FftPlan(4096, 4, sd.forward_fft, sd.F32, max_batch=batch)
    rev = sd.FftPlan(4096, 4, sd.reverse_fft, sd.F32, max_batch=batch)
    variants = [int(v) for v in args.variants.split(",")] if args.variants else [0, 1, 2]
    bytes_per = batch * 65536

    def run(v):
        fwd.set_variant(v)
        rev.set_variant(v)
        fwd.exec(x)
        rev.exec(x)
    launches = 2
else:
    ch = args.units or (1 << 20)
    f64 = args.what == "iir64"
    x = torch.randn((ch, 4096), device=dev, dtype=torch.float64 if f64 else torch.float32)
    bank = sd.casc_2o_iir(4, ch, sd.F64 if f64 else sd.F32)
    bank.set_lp_coeff(10e3, 100e3)
    variants = [int(v) for v in args.variants.split(",")] if args.variants else [0, 1, 2]
    bytes_per = ch * 4096 * (16 if f64 else 8)

    def run(v):
        bank.set_variant(v)
        bank.reset()
        bank.process(x)
    launches = 1

times = {v: [] for v in variants}
for v in variants:
    run(v)
torch.cuda.synchronize()
for r in range(args.rounds):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(v)
        e1.record()
        e1.synchronize()
        times[v].append(e0.elapsed_time(e1) / launches)
for v in variants:
    med, mn = statistics.median(times[v]), min(times[v])
    print(f"variant {v}: median {med:.4f} ms  min {mn:.4f} ms  -> {bytes_per / med / 1e6:8.1f} GB/s median, "
          f"{bytes_per / mn / 1e6:8.1f} GB/s best  ({100 * bytes_per / med / 1e6 / 8000:.1f} % of 8 TB/s)")
