#!/usr/bin/env python3
"""Time of one exec against the batch size, persistent launch (p) and two launches per chunk (c): where the fixed cost is.
tools/lab_fft2p_batch.py [log2 n = 17] [f32|f64]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
f64 = len(sys.argv) > 2 and sys.argv[2] == "f64"
prec = sd.F64 if f64 else sd.F32
n, es = 1 << k, 16 if f64 else 8
probe = sd.FftPlan(n, 2, sd.forward_fft, prec, max_batch=max(16, (1 << 28) // (n * es)))
fused = 0 if probe.info.kernel.decode() == "sdsp_fft2p_fused" else 3
for mib in (256, 512, 1024, 2048, 4096, 8192):
    batch = (mib << 20) // (n * es)
    x = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda", dtype=torch.float64 if f64 else torch.float32))
    for name, variant in (("p", fused), ("c", 3 - fused)):
        fwd = sd.FftPlan(n, 2, sd.forward_fft, prec, max_batch=batch); rev = sd.FftPlan(n, 2, sd.reverse_fft, prec, max_batch=batch)
        fwd.set_variant(variant); rev.set_variant(variant)
        for _ in range(2):
            fwd.exec(x); rev.exec(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fwd.exec(x); rev.exec(x)
        e1.record(); torch.cuda.synchronize()
        fwd.status()
        ms = e0.elapsed_time(e1) / 8
        print(f"N=2^{k} {'f64' if f64 else 'f32'} {mib:5d} MiB {name} [{fwd.info.kernel.decode()}, {fwd.launches(batch)} launches]: {ms:7.3f} ms, "
              f"{2*(mib<<20)/ms/1e6/80:.1f} %", flush=True)
        del fwd, rev
    del x
