#!/usr/bin/env python3
"""f64 throughput per size and kernel variant (1 GiB of complex128, forward / reverse alternating, same-call numbers):
tools/sweep_sizes64.py [rounds] [radix = 2]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
radix = int(sys.argv[2]) if len(sys.argv) > 2 else 2
total = 1 << 26  # complex128 elements = 1 GiB
buf = torch.view_as_complex(torch.randn((total, 2), device="cuda", dtype=torch.float64))
for rep in range(rounds):
    for n, variants in ((64, (0,)), (1024, (0,)), (4096, (0, 1)), (8192, (0, 1)), (16384, (0, 1)), (1 << 15, (0, 1, 3)), (1 << 16, (0, 3)), (1 << 20, (0, 3))):
        if radix == 4 and not sd.isPowerOf4(n):
            continue
        batch = total // n
        x = buf.view(batch, n)
        for variant in variants:
            fwd = sd.FftPlan(n, radix, sd.forward_fft, sd.F64, max_batch=min(batch, 4096)); rev = sd.FftPlan(n, radix, sd.reverse_fft, sd.F64, max_batch=min(batch, 4096))
            fwd.set_variant(variant); rev.set_variant(variant)
            for _ in range(2):
                fwd.exec(x); rev.exec(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fwd.exec(x); rev.exec(x)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 8
            print(f"round {rep} N={n:7d} f64 radix {radix} variant {variant} [{fwd.info.kernel.decode()}, {fwd.info.hbm_passes} pass]: {ms:7.3f} ms per GiB -> "
                  f"{batch/ms/1e3:8.2f} M FFT/s, {2*total*16/ms/1e6/80:.1f} % of 8 TB/s", flush=True)
            del fwd, rev
    buf.normal_()
