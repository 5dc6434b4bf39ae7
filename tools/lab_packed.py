#!/usr/bin/env python3
"""One-call A/B of fft_big.hip's packed-arithmetic form (plan variant 4) against the scalar form (variant 5): plain transform and
fused convolution, N = 8192 / 16384 / 32768, 1 GiB batches, interleaved, `rounds` times.  tools/lab_packed.py [rounds]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import simpledsp_amd as sd

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
total = 1 << 27
buf = torch.view_as_complex(torch.randn((total, 2), device="cuda"))
# parity of the two forms against each other and numpy (a few transforms)
for n in (8192, 16384, 32768):
    x = (np.random.default_rng(n).standard_normal((3, n)) + 1j * np.random.default_rng(n + 1).standard_normal((3, n))).astype(np.complex64)
    want = np.fft.fft(x.astype(np.complex128), axis=-1)
    for v in (5, 4):
        p = sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=4)
        p.set_variant(v)
        d = torch.from_numpy(x).cuda()
        p.exec(d)
        torch.cuda.synchronize()
        err = np.abs(d.cpu().numpy() - want).max(axis=1) / np.abs(want).max(axis=1)
        print(f"N={n} variant {v} [{p.info.kernel.decode()}]: rel err {err.max():.2e}", flush=True)
for rep in range(rounds):
    for n in (8192, 16384, 32768):
        batch = total // n
        x = buf.view(batch, n)
        ph = torch.rand((n,), device="cuda") * 6.283185307179586
        h = torch.polar(torch.ones_like(ph), ph)
        for v in (5, 4):
            fwd = sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=64); rev = sd.FftPlan(n, 2, sd.reverse_fft, sd.F32, max_batch=64)
            fwd.set_variant(v); rev.set_variant(v)
            for _ in range(6):
                fwd.exec(x); rev.exec(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fwd.exec(x); rev.exec(x)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            for _ in range(4):
                fwd.convolve(x, h)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                fwd.convolve(x, h)
            e1.record(); torch.cuda.synchronize()
            msc = e0.elapsed_time(e1) / 10
            print(f"round {rep} N={n:6d} {'packed' if v == 4 else 'scalar'}: transform {2*total*8/ms/1e6/80:5.1f} %   fused convolution {2*total*8/msc/1e6/80:5.1f} % of 8 TB/s", flush=True)
    buf.normal_()
