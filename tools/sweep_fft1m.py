#!/usr/bin/env python3
"""Chunk-size sweep of the N=2^20 path (variant -> transforms per launch pair), sustained timing."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

batch = 256
n = 1 << 20
dev = torch.device("cuda:0")
x = torch.view_as_complex(torch.randn((batch, n, 2), device=dev))
fwd = sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=batch)
rev = sd.FftPlan(n, 2, sd.reverse_fft, sd.F32, max_batch=batch)
chunk_of = {0: "32", 1: "16 overlap", 2: "16", 3: "8 overlap", 4: "24", 5: "12 overlap", 6: "8", 7: "4 overlap", 99: "generic"}
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 3, 5, 7, 2]
for v in variants:
    fwd.set_variant(v); rev.set_variant(v)
    for _ in range(2):
        fwd.exec(x); rev.exec(x)
    torch.cuda.synchronize()
    reps = 6 if v != 99 else 2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fwd.exec(x); rev.exec(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (2 * reps)
    print(f"variant {v} (chunk {chunk_of[v]}): {ms:.3f} ms per 256 transforms -> {batch/ms*1e3:.0f} FFT/s, "
          f"compulsory {batch*16*2**20/ms/1e6:.0f} GB/s = {batch*16*2**20/ms/1e6/80:.1f} % of 8 TB/s")
