import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch, simpledsp_amd as sd
n, batch = 1 << 23, 8
x = torch.view_as_complex(torch.randn((batch, n, 2), device="cuda"))
want = np.fft.fft(x[[0, 7]].cpu().numpy().astype(np.complex128), axis=-1)
p = sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=batch)
print(p.info.kernel.decode(), p.info.hbm_passes, p.launches(batch))
y = x.clone(); p.exec(y); p.status()
got = y[[0, 7]].cpu().numpy()
print("rel err", np.abs(got - want).max() / np.abs(want).max())
r = sd.FftPlan(n, 2, sd.reverse_fft, sd.F32, max_batch=batch)
for _ in range(2): p.exec(y); r.exec(y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): p.exec(y); r.exec(y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"{ms:.3f} ms per {batch} transforms: {2*batch*n*8/ms/1e6/80:.1f} % of HBM peak")
