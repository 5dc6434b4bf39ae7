#!/usr/bin/env python3
"""Fast convolution at the two-pass sizes: fused multiply (the forward transform's pass 2 multiplies by h on its way out: 4 passes over HBM)
against forward + multiply + reverse as separate steps through the same plans (5 passes); 1 GiB batches; % of HBM peak on the compulsory bytes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd
for prec, n in [(sd.F32, 1 << k) for k in (16, 18, 20, 22)] + [(sd.F64, 1 << k) for k in (16, 20)]:
    f64 = prec == sd.F64
    batch = ((1 << 26) if f64 else (1 << 27)) // n
    rdt = torch.float64 if f64 else torch.float32
    x = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda", dtype=rdt))
    ph = torch.rand((n,), device="cuda", dtype=rdt) * 6.283185307179586
    h = torch.polar(torch.ones_like(ph), ph)
    fwd = sd.FftPlan(n, 2, sd.forward_fft, prec, max_batch=batch); rev = sd.FftPlan(n, 2, sd.reverse_fft, prec, max_batch=batch)
    def fused():
        fwd.convolve(x, h)
    def separate():
        fwd.exec(x); x.mul_(h); rev.exec(x)
    for name, fn in (("convolve (fused multiply)", fused), ("exec, multiply, exec", separate)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 6
        print(f"N=2^{n.bit_length()-1} {'f64' if f64 else 'f32'} {name:28s}: {ms:.3f} ms per GiB, {2*(1<<30)/ms/1e6/80:.1f} % of HBM peak", flush=True)
