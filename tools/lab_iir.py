#!/usr/bin/env python3
"""One-call A/B of the biquad bank's kernel variants on BASELINE configs[3] (1M channels x 4096 samples):
    tools/lab_iir.py [f32|f64|mix] V0,V1,...  [channels]  [rounds] [check]
Variants (csrc/iir.hip: iir_select): 0 default (f32: landing slot), 1 wide super-tile, 2 direct, 3 super-tile; lab, f32 only:
10 LDS-DMA ring, 19 landing slot without the recurrence, 20 ring without the recurrence.  (profiles/r03_iir_lab.md's tables were
made with the lab numbering of the time: 18 = today's 0, 0 = today's 3, 9 / 11-17 / 21-29 = ring shapes no longer instantiated.)
Every variant is first checked bit-for-bit against variant 0 on a small bank (the kernels share cascade_step, so any
difference is an addressing bug), then all variants are timed interleaved in one process, `rounds` times (same-call
numbers: the only ones that may be compared to a point)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

prec_name = sys.argv[1] if len(sys.argv) > 1 else "f32"
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
channels = int(sys.argv[3]) if len(sys.argv) > 3 else (1 << 19 if prec_name == "f64" else 1 << 20)
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
prec = {"f32": sd.F32, "f64": sd.F64, "mix": sd.F32_F64STATE}[prec_name]
dt = torch.float64 if prec_name == "f64" else torch.float32
samples = 4096

# ---- parity: each variant against variant 0, two consecutive blocks (state carried over), 3 x 64 channels
g = torch.Generator(device="cuda").manual_seed(7)
small = torch.randn((192, 2048), generator=g, device="cuda", dtype=dt)
ref = None
for v in [0] + [v for v in variants if v != 0]:
    bank = sd.casc_2o_iir(4, 192, prec, sd.IIR_GENERIC)
    bank.set_lp_coeff(10e3, 100e3)
    bank.set_variant(v)
    y = small.clone()
    bank.process(y, 1024, 0)
    bank.process(y, 1024, 1024)
    torch.cuda.synchronize()
    if v == 0:
        ref = y
    else:
        same = torch.equal(y, ref) or v >= 19  # 19 ..: copy-only lab variants
        print(f"variant {v}: {'bit-identical to variant 0' if torch.equal(y, ref) else 'DIFFERENT from variant 0'}", flush=True)
        if not same:
            bad = (y != ref).nonzero()
            print("  first differences (channel, sample):", bad[:8].tolist(), flush=True)
            sys.exit(1)

# ---- timing
x = torch.randn((channels, samples), generator=g, device="cuda", dtype=dt)
# full-size parity first (the small case cannot show an ordering bug that needs a loaded memory system)
if len(sys.argv) > 5 and sys.argv[5] == "check":
    keep = x.clone()
    want = None
    for v in [0] + [v for v in variants if 0 < v < 19]:
        b = sd.casc_2o_iir(4, channels, prec, sd.IIR_GENERIC)
        b.set_lp_coeff(10e3, 100e3)
        b.set_variant(v)
        x.copy_(keep)
        b.process(x)
        torch.cuda.synchronize()
        if v == 0:
            want = x.clone()
        else:
            print(f"full size, variant {v}: {'bit-identical' if torch.equal(x, want) else 'DIFFERENT'}", flush=True)
    del keep, want
    x.normal_()
banks = {}
for v in variants:
    b = sd.casc_2o_iir(4, channels, prec, sd.IIR_GENERIC)
    b.set_lp_coeff(10e3, 100e3)
    b.set_variant(v)
    banks[v] = b
unit = 16 if prec_name == "f64" else 8
for rep in range(rounds):
    for v in variants:
        b = banks[v]
        for _ in range(3):
            b.reset(); b.process(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            b.reset(); b.process(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        print(f"round {rep} {prec_name} variant {v}: {ms:.3f} ms, {channels*samples*unit/ms/1e6/80:.2f} % of 8 TB/s", flush=True)
    x.normal_()
