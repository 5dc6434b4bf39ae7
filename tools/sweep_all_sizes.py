#!/usr/bin/env python3
"""Every power of two 16 ... 2^22 (f32) and 16 ... 2^20 (f64) through the plan's default kernel, radix 2 and (where N is a power of 4) radix 4:
1 GiB batches, forward / reverse alternating, 12 untimed + 10 timed pairs; one call = one table.  % of HBM peak on the compulsory bytes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

for prec, name, top, es in ((sd.F32, "f32", 22, 8), (sd.F64, "f64", 20, 16)):
    total = (1 << 30) // es
    buf = torch.view_as_complex(torch.randn((total, 2), device="cuda", dtype=torch.float64 if prec == sd.F64 else torch.float32))
    print(f"| N ({name}) | radix 2 | kernel | radix 4 | kernel |\n|---|---|---|---|---|")
    for k in range(4, top + 1):
        n = 1 << k
        cells = []
        for radix in (2, 4):
            if radix == 4 and k % 2:
                cells += ["—", ""]
                continue
            batch = total // n
            x = buf.view(batch, n)
            cap = batch if n >= 32768 else min(batch, 64)
            fwd = sd.FftPlan(n, radix, sd.forward_fft, prec, max_batch=cap); rev = sd.FftPlan(n, radix, sd.reverse_fft, prec, max_batch=cap)
            for _ in range(12):
                fwd.exec(x); rev.exec(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fwd.exec(x); rev.exec(x)
            e1.record(); torch.cuda.synchronize()
            fwd.status()
            ms = e0.elapsed_time(e1) / 20
            cells += [f"{2 * (1 << 30) / ms / 1e6 / 80:.1f} %", f"`{fwd.info.kernel.decode()}` ({fwd.info.hbm_passes})"]
            del fwd, rev
        print(f"| 2^{k} = {n} | " + " | ".join(cells) + " |", flush=True)
        buf.normal_()
    del buf
