#!/usr/bin/env python3
"""Soak of the persistent two-pass launch: every two-pass size, 2 GiB batches, the same input transformed R times by the persistent
launch (variant 0) and compared bit for bit with the two launches per chunk (variant 3); forward and reverse; status() after each.
tools/soak_fft2p_fused.py [R = 6]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cases = [(sd.F32, 1 << k) for k in (16, 17, 18, 19, 21, 22)] + [(sd.F64, 1 << k) for k in range(15, 21)]
bad = 0
for prec, n in cases:
    f64 = prec == sd.F64
    batch = (1 << 31) // (n * (16 if f64 else 8)) - 3  # ragged: a short last unit
    g = torch.Generator(device="cuda").manual_seed(n % 4093 + (5 if f64 else 0))
    x0 = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda", dtype=torch.float64 if f64 else torch.float32, generator=g))
    for T in (sd.forward_fft, sd.reverse_fft):
        plan = sd.FftPlan(n, 2, T, prec, max_batch=batch)
        assert plan.info.kernel.decode() == "sdsp_fft2p_fused", plan.info.kernel
        plan.set_variant(3)
        ref = x0.clone(); plan.exec(ref); plan.status()
        plan.set_variant(0)
        for r in range(R):
            y = x0.clone(); plan.exec(y); plan.status()
            if not torch.equal(torch.view_as_real(y), torch.view_as_real(ref)):
                bad += 1
                d = (torch.view_as_real(y) != torch.view_as_real(ref)).any(dim=-1)
                print(f"MISMATCH N=2^{n.bit_length()-1} {'f64' if f64 else 'f32'} dir {T} rep {r}: {int(d.sum())} elements, first transform {int(d.any(dim=1).nonzero()[0])}", flush=True)
        del plan, ref
    print(f"N=2^{n.bit_length()-1} {'f64' if f64 else 'f32'} batch {batch}: {2 * R} persistent launches checked", flush=True)
    del x0
print("mismatches:", bad)
sys.exit(1 if bad else 0)
