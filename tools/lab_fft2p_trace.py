#!/usr/bin/env python3
"""LAB (needs the SDSP_HIP_LAB_TRACE hook compiled in): when does each unit's pass 1 / pass 2 complete inside one persistent launch?"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
n = 1 << k
batch = (mib << 20) // (n * 8)
unit = max(1, (8 << 20) // (n * 8))
units = (batch + unit - 1) // unit
x = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda"))
trace = torch.zeros(2 * units, dtype=torch.int64, device="cuda")
fwd = sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=batch); rev = sd.FftPlan(n, 2, sd.reverse_fft, sd.F32, max_batch=batch)
fused = 0 if fwd.info.kernel.decode() == "sdsp_fft2p_fused" else 3
fwd.set_variant(fused); rev.set_variant(fused)
for _ in range(3):
    fwd.exec(x); rev.exec(x)
torch.cuda.synchronize()
os.environ["SDSP_HIP_LAB_TRACE"] = hex(trace.data_ptr())
fwd.exec(x)
torch.cuda.synchronize()
os.environ.pop("SDSP_HIP_LAB_TRACE")
t = trace.cpu().numpy().reshape(units, 2).astype("float64")
t0 = t[t > 0].min()
t = (t - t0) / 100.0  # us (100 MHz)
print(f"N=2^{k}, {mib} MiB, {units} units of {unit} transforms; unit id = queue + 8 * index; times in us since the first completion")
for q in range(8):
    idx = list(range(q, units, 8))
    p1 = " ".join(f"{t[u,0]:6.0f}" for u in idx[:40])
    p2 = " ".join(f"{t[u,1]:6.0f}" for u in idx[:40])
    print(f"queue {q} pass 1 done: {p1}")
    print(f"queue {q} pass 2 done: {p2}")
print("last completion per queue:", [round(t[list(range(q, units, 8)), 1].max()) for q in range(8)])
