#!/usr/bin/env python3
"""Lab (needs tools/lab_patches/fft_big_lab.patch applied to csrc/fft_big.hip): sdsp_fft_big_kernel under the SDSP_LAB_BIG knobs -- compute-only, memory-only and a
staggered start of the first round of workgroups.  SDSP_LAB_BIG = "lab,first,groups,ticks" (ticks of 10 ns)."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

dev = torch.device("cuda:0")
gib = float(os.environ.get("LAB_GIB", "1"))
total = int((1 << 27) * gib)
buf = torch.view_as_complex(torch.randn((total, 2), device=dev))


def run(n, knobs, radix=2, reps=10):
    os.environ["SDSP_LAB_BIG"] = knobs
    batch = total // n
    x = buf.view(batch, n)
    fwd = sd.FftPlan(n, radix, sd.forward_fft, sd.F32, max_batch=min(batch, 64))
    rev = sd.FftPlan(n, radix, sd.reverse_fft, sd.F32, max_batch=min(batch, 64))
    for _ in range(12):
        fwd.exec(x); rev.exec(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fwd.exec(x); rev.exec(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (2 * reps)
    us_per = ms * 1e3 / (batch / 256.0)  # per transform per CU
    print(f"N={n:6d} knobs {knobs:>18s}: {ms:7.3f} ms per {gib:g} GiB = {2*total*8/ms/1e6/80:5.1f} %  ({us_per:6.2f} us per transform per CU)"
          f"  {fwd.info.kernel.decode()}", flush=True)
    buf.normal_()


for n in (8192, 16384, 32768):
    run(n, "0,0,0,0,0,0")
    run(n, "0,0,0,0,1,0")
    run(n, "1,0,0,0,0,0")
    run(n, "1,0,0,0,1,0")
    run(n, "2,0,0,0,0,0")
    run(n, "2,0,0,0,1,0")
    run(n, "0,0,0,0,0,0")
    run(n, "0,0,0,0,1,0")
# correctness of the persistent loop against the shipped kernel (bitwise: same arithmetic)
import numpy as np
for n in (16384, 32768):
    for batch in (1, 300, 1000):
        x0 = torch.view_as_complex(torch.randn((batch, n, 2), device=dev))
        outs = []
        for knobs in ("0,0,0,0,0,0", "0,0,0,0,1,0"):
            os.environ["SDSP_LAB_BIG"] = knobs
            x = x0.clone()
            sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=batch).exec(x)
            torch.cuda.synchronize()
            outs.append(x)
        ref = torch.fft.fft(x0.to(torch.complex128))
        err = ((outs[1] - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max().item()
        print(f"N={n} batch {batch}: persistent == shipped bitwise: {torch.equal(outs[0], outs[1])}; rel err vs torch f64 {err:.2e}", flush=True)
