#!/usr/bin/env python3
"""tools/iir_stride_probe.py -- does the power-of-two row pitch of BASELINE config 4 (4096 floats = 16 KiB) cost the IIR kernel
anything?  Every wave walks its 64 rows in the same order, so at any moment the chip asks for the same column window of
131 072 rows 16 KiB apart; if the memory channels were picked from low address bits alone that would be a hot spot.  Times
the default kernel on rows of 4096 samples with a pitch of 4096 + pad floats (only the 4096 samples are processed)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import simpledsp_amd as sd

channels = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
samples = 4096
dev = torch.device("cuda", 0)
sd.load()
for pad in (0, 32, 64, 128, 256, 1024, 0):
    x = torch.randn((channels, samples + pad), device=dev, dtype=torch.float32)
    bank = sd.casc_2o_iir(4, channels, sd.F32, sd.IIR_GENERIC)
    bank.set_lp_coeff(10e3, 100e3, 1.0)
    for _ in range(5):
        bank.process(x, samples=samples, offset=0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        bank.process(x, samples=samples, offset=0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"pitch {samples + pad:5d} floats: {ms:.3f} ms per pass, {channels * samples * 8 / (ms * 1e-3) / 8e12 * 100:.2f} % of HBM peak", flush=True)
    del x, bank
    torch.cuda.empty_cache()
