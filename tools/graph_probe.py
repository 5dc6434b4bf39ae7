#!/usr/bin/env python3
"""Launch-bound loops, eager against captured into ONE graph and replayed (the library only enqueues: include/sdsp_hip.h):
(a) block streaming through a biquad bank -- 4096 channels x 4096 samples in 32 blocks of 128 (testIIR.cpp:61-75's pattern, state in the bank);
(b) 64 small FFT calls back to back (N = 1024, batch 256 each, forward / reverse alternating)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import simpledsp_amd as sd


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    return g


bank = sd.casc_2o_iir(4, 4096, sd.F32, sd.IIR_GENERIC)
bank.set_lp_coeff(10e3, 100e3)
d = torch.randn((4096, 4096), device="cuda")
def blocks():
    for off in range(0, 4096, 128):
        bank.process(d, samples=128, offset=off)
blocks(); torch.cuda.synchronize()
e = timed(blocks)
g = capture(blocks)
r = timed(g.replay)
print(f"(a) biquad bank, 32 blocks of 128 samples x 4096 channels: eager {e:.0f} us per pass, graph replay {r:.0f} us ({e / r:.2f}x)")

fwd = sd.FftPlan(1024, 2, sd.forward_fft, sd.F32, max_batch=256); rev = sd.FftPlan(1024, 2, sd.reverse_fft, sd.F32, max_batch=256)
x = torch.view_as_complex(torch.randn((256, 1024, 2), device="cuda"))
def ffts():
    for _ in range(32):
        fwd.exec(x); rev.exec(x)
ffts(); torch.cuda.synchronize()
e = timed(ffts)
g = capture(ffts)
r = timed(g.replay)
print(f"(b) 64 FFT calls of N = 1024, batch 256: eager {e:.0f} us, graph replay {r:.0f} us ({e / r:.2f}x)")
