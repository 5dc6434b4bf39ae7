#!/usr/bin/env python3
"""Throughput of every (N, radix) through whatever kernel the plan picks, f32, 1 GiB batches."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

dev = torch.device("cuda:0")
total = 1 << 27  # complex elements = 1 GiB
buf = torch.view_as_complex(torch.randn((total, 2), device=dev))
sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
for n in sizes:
    for radix in (2, 4, 0):  # 0 = SDSP_HIP_RADIX_AUTO: the fastest kernel of the size (mixed radix at N = 2 * 4^k)
        if radix == 4 and not sd.isPowerOf4(n):
            continue
        if radix == 0 and n not in (8192, 16384):
            continue  # elsewhere AUTO is one of the two above
        batch = total // n
        x = buf.view(batch, n)
        cap = batch if n >= 65536 else min(batch, 64)  # multi-pass sizes chunk by the plan's workspace: give it the whole batch
        fwd = sd.FftPlan(n, radix, sd.forward_fft, sd.F32, max_batch=cap)
        rev = sd.FftPlan(n, radix, sd.reverse_fft, sd.F32, max_batch=cap)
        for _ in range(12):  # steady state: the first ~20 launches after idle run slower (clock ramp)
            fwd.exec(x); rev.exec(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            fwd.exec(x); rev.exec(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (2 * reps)
        print(f"N={n:6d} radix {radix}: {ms:8.3f} ms per GiB -> {batch/ms*1e3/1e6:9.2f} M FFT/s, {2*total*8/ms/1e6:7.0f} GB/s "
              f"({2*total*8/ms/1e6/80:.1f} %)  kernel {fwd.info.kernel.decode()}")
