#!/usr/bin/env python3
"""Schedule knobs of the persistent two-pass launch (LAB hook SDSP_HIP_LAB_F2 = unit,queues,ring,lag,block), same call:
tools/lab_fft2p_fused_cfg.py [rounds]"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = [(sd.F32, 1 << k) for k in (16, 17, 18, 19, 21, 22)] + [(sd.F64, 1 << k) for k in (16, 18, 20)]
for rep in range(rounds):
    for prec, n in cases:
        f64 = prec == sd.F64
        total = (1 << 26) if f64 else (1 << 27)  # 1 GiB
        batch = total // n
        x = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda", dtype=torch.float64 if f64 else torch.float32))
        u0 = max(1, (8 << 20) // (n * (16 if f64 else 8)))
        ub = u0 * n * (16 if f64 else 8)
        q0 = max(2, (256 << 20) // (ub * 4))
        cfgs = [None] + [(u0, q0, 4, 2, b) for b in (1, 2, 4, 8)] + [(u0, q0, 3, 1, 2)]
        if u0 >= 4:
            cfgs += [(u0 // 4, q0 * 4, 4, 2, b) for b in (1, 2, 4)]
        if u0 >= 2:
            cfgs += [(u0 // 2, q0 * 2, 4, 2, b) for b in (2, 4)]
        if q0 >= 4:
            cfgs += [(u0, q0 // 2, 4, 2, 2), (u0, q0 // 2, 8, 2, 2)]
        for cfg in cfgs:
            os.environ.pop("SDSP_HIP_LAB_F2", None)
            if cfg:
                os.environ["SDSP_HIP_LAB_F2"] = ",".join(str(v) for v in cfg)
            fwd = sd.FftPlan(n, 2, sd.forward_fft, prec, max_batch=batch); rev = sd.FftPlan(n, 2, sd.reverse_fft, prec, max_batch=batch)
            fwd.set_variant(3 if cfg else 0); rev.set_variant(3 if cfg else 0)
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
            print(f"round {rep} N=2^{n.bit_length()-1} {'f64' if f64 else 'f32'} {fwd.info.kernel.decode():32s} unit,queues,ring,lag,block = {cfg}: "
                  f"{ms:7.3f} ms per GiB, {2*(1<<30)/ms/1e6/80:.1f} %", flush=True)
            del fwd, rev
        assert torch.isfinite(torch.view_as_real(x)).all()
        del x
