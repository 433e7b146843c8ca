import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for Q in (200, 400):
    q = torch.randn(1, Q, 128, device='cuda').half(); kv = torch.randn(1, 32400, 256, device='cuda').half()
    print(f'mha_decode fp16 Q={Q}: {t(lambda: ops.mha_decode(q, kv, 8, 0.25)):.1f} us')
    print(f'mha_decode fp32 Q={Q}: {t(lambda: ops.mha_decode(q.float(), kv.float(), 8, 0.25)):.1f} us')
