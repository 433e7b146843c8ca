"""Calibration: achievable read / copy bandwidth of this GPU with plain torch kernels (MB-sized and GB-sized buffers)."""
import torch
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e-3
for mb in (34, 103, 400, 2000):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device='cuda'); y = torch.empty_like(x)
    dt = t(lambda: y.copy_(x)); print(f'copy  {mb:5d} MB: {2 * n * 4 / dt / 1e12:.2f} TB/s (read+write)  {dt*1e6:.1f} us')
    dt = t(lambda: x.sum());     print(f'sum   {mb:5d} MB: {n * 4 / dt / 1e12:.2f} TB/s (read)')
    dt = t(lambda: y.zero_());   print(f'zero  {mb:5d} MB: {n * 4 / dt / 1e12:.2f} TB/s (write)')
