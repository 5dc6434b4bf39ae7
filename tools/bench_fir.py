#!/usr/bin/env python3
"""FIR bank throughput on the BASELINE config-4 shape: channels x 4096 samples, in place."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

precision = sys.argv[1] if len(sys.argv) > 1 else "f32"
taps_list = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16, 32, 64, 128]
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1]
f64 = precision == "f64"
channels, samples = (1 << 19) if f64 else (1 << 20), 4096
x = torch.randn((channels, samples), device="cuda", dtype=torch.float64 if f64 else torch.float32)
rs = 8 if f64 else 4
for taps in taps_list:
    for variant in variants:
        bank = sd.fir_filter(taps, channels, sd.F64 if f64 else sd.F32)
        bank.set_lp_coeff(10e3, 100e3)  # unit DC gain: repeated filtering stays bounded
        bank.set_variant(variant)
        for _ in range(10):
            bank.process(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            bank.process(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gbs = channels * samples * rs * 2 / ms / 1e6
        print(f"{precision} taps {taps:4d} variant {variant}: {ms:.3f} ms -> {channels*samples/ms/1e9:.3f} T samples/s, "
              f"{gbs:.0f} GB/s = {gbs/80:.1f} % of 8 TB/s, {2*taps*channels*samples/ms/1e9:.1f} TFLOP/s", flush=True)
