#!/usr/bin/env python3
"""Sustained (back-to-back) launch timing per variant: chunks of 20 launches between HIP events,
no host sync inside a variant's run.  Shows the DVFS/load transient that short sweeps hide."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2]
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
batch = 65536
dev = torch.device("cuda:0")
x = torch.view_as_complex(torch.randn((batch, 4096, 2), device=dev))
fwd = sd.FftPlan(4096, 4, sd.forward_fft, sd.F32, max_batch=batch)
rev = sd.FftPlan(4096, 4, sd.reverse_fft, sd.F32, max_batch=batch)
for v in variants:
    fwd.set_variant(v); rev.set_variant(v)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    evs[0].record()
    for c in range(chunks):
        for i in range(10):
            fwd.exec(x); rev.exec(x)
        evs[c + 1].record()
    torch.cuda.synchronize()
    ms = [evs[c].elapsed_time(evs[c + 1]) / 20 for c in range(chunks)]
    gb = [batch * 65536 / m / 1e6 for m in ms]
    print(f"variant {v}: ms/launch per chunk of 20: " + " ".join(f"{m:.3f}" for m in ms))
    print(f"           GB/s: " + " ".join(f"{g:.0f}" for g in gb) + f"   | last-half mean {sum(gb[chunks//2:])/len(gb[chunks//2:]):.0f} GB/s")
