#!/usr/bin/env python3
"""Chunk size of the two-launch two-pass schedule (LAB hook SDSP_HIP_LAB_CHUNK = MiB of intermediate per chunk), same call."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    for prec, n in [(sd.F32, 1 << k) for k in (16, 19, 21, 22)] + [(sd.F64, 1 << k) for k in (16, 20)]:
        f64 = prec == sd.F64
        total = (1 << 27) if f64 else (1 << 28)  # 2 GiB
        batch = total // n
        x = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda", dtype=torch.float64 if f64 else torch.float32))
        for mib in (160, 192, 224, 256, 288, 320, 384):
            if mib < n * (16 if f64 else 8) >> 20:
                continue
            os.environ["SDSP_HIP_LAB_CHUNK"] = str(mib)
            fwd = sd.FftPlan(n, 2, sd.forward_fft, prec, max_batch=batch); rev = sd.FftPlan(n, 2, sd.reverse_fft, prec, max_batch=batch)
            for _ in range(2):
                fwd.exec(x); rev.exec(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fwd.exec(x); rev.exec(x)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 6
            print(f"round {rep} N=2^{n.bit_length()-1} {'f64' if f64 else 'f32'} chunk {mib:5d} MiB, {fwd.launches(batch):3d} launches: "
                  f"{ms:7.3f} ms per 2 GiB, {4*(1<<30)/ms/1e6/80:.1f} %", flush=True)
            del fwd, rev
        del x
