#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry (sdsp_hip_fft_exec_host): H2D + kernel + D2H.
Never bench.py's `value`; recorded in DESIGN.md section 6."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import simpledsp_amd as sd

batch = 16384  # 512 MiB each way
x = (np.random.default_rng(0).standard_normal((batch, 4096, 2)).astype(np.float32)).view(np.complex64)[..., 0]
plan = sd.FftPlan(4096, 4, sd.forward_fft, sd.F32, max_batch=batch)
plan.exec_host(x)  # warm: staging buffer allocation
ts = []
for _ in range(3):
    t0 = time.perf_counter(); plan.exec_host(x); ts.append(time.perf_counter() - t0)
t = min(ts)
print(f"exec_host: {batch} transforms in {t*1e3:.1f} ms -> {batch/t/1e6:.2f} M FFT/s, {2*x.nbytes/t/1e9:.1f} GB/s over PCIe (pageable host memory)")
