#!/usr/bin/env python3
"""The two-pass sizes in one persistent launch (variant A) against two launches per chunk (variant B), same call:
results must be bit-identical (same tiles, same arithmetic); 1 GiB of data, forward / reverse alternating.
tools/lab_fft2p_fused.py [rounds] [A] [B] [GiB] [cases]      (defaults: 2 rounds, variants 3 and 0, 1 GiB, every two-pass size)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
va = int(sys.argv[2]) if len(sys.argv) > 2 else 3
vb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
gib = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cases = [(sd.F32, n) for n in (1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 21, 1 << 22)] + [(sd.F64, 1 << k) for k in (16, 17, 18, 20)]
if len(sys.argv) > 5:  # e.g. f64:18,f64:20,f32:22
    cases = [(sd.F64 if c.split(":")[0] == "f64" else sd.F32, 1 << int(c.split(":")[1])) for c in sys.argv[5].split(",")]
for rep in range(rounds):
    for prec, n in cases:
        f64 = prec == sd.F64
        total = gib * ((1 << 26) if f64 else (1 << 27))
        batch = total // n
        gen = torch.Generator(device="cuda").manual_seed(1000 * rep + n % 1000 + (7 if f64 else 0))
        x0 = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda", dtype=torch.float64 if f64 else torch.float32, generator=gen))
        out = {}
        for variant in (va, vb):
            fwd = sd.FftPlan(n, 2, sd.forward_fft, prec, max_batch=batch); rev = sd.FftPlan(n, 2, sd.reverse_fft, prec, max_batch=batch)
            fwd.set_variant(variant); rev.set_variant(variant)
            x = x0.clone()
            fwd.exec(x)
            f1 = x.clone()
            rev.exec(x)
            torch.cuda.synchronize()
            fwd.status(); rev.status()
            out[variant] = (f1, x.clone())
            for _ in range(2):
                fwd.exec(x); rev.exec(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fwd.exec(x); rev.exec(x)
            e1.record(); torch.cuda.synchronize()
            fwd.status(); rev.status()
            ms = e0.elapsed_time(e1) / 8
            print(f"round {rep} N=2^{n.bit_length()-1} {'f64' if f64 else 'f32'} variant {variant} [{fwd.info.kernel.decode()}, {fwd.launches(batch)} launches]: "
                  f"{ms:7.3f} ms per {gib} GiB, {2*gib*(1<<30)/ms/1e6/80:.1f} % of 8 TB/s", flush=True)
            del fwd, rev, x
        same = all(torch.equal(torch.view_as_real(a), torch.view_as_real(b)) for a, b in zip(out[va], out[vb]))
        rt = (out[va][1] - x0).abs().max().item()
        print(f"        bit-identical to variant {vb}: {same}; round trip max |err| {rt:.3g}", flush=True)
        assert same
        del out, x0
