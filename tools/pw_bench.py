"""Timing of the fused 1x1-conv chains against the library-GEMM form, image-side shape."""
import os, sys, math
import torch
import torch.nn.functional as F
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
for name, (n, H, W) in dict(image=(6, 112, 200), bev=(1, 180, 180)).items():
    g = torch.Generator(device='cuda').manual_seed(0)
    mk = lambda: torch.randn(n, 128, H, W, device='cuda', generator=g).half().contiguous(memory_format=torch.channels_last)
    x1, x2, x3 = mk(), mk(), mk()
    w = lambda k: (torch.randn(128, k, device='cuda', generator=g) / math.sqrt(k)).half()
    b = torch.randn(128, device='cuda', generator=g)
    w1, w2, wa, wb = w(128), w(128), w(256), w(256)
    M = n * H * W
    print(name, 'v-proj  (128)      fused %.1f us' % t(lambda: ops.pointwise_chain(x1, w1, b, True)))
    print(name, 'q-chain (128,128)  fused %.1f us' % t(lambda: ops.pointwise_chain(x1, w1, b, True, w2=w2, b2=b, relu2=True)))
    print(name, 'mix2    (256,256)  fused %.1f us' % t(lambda: ops.pointwise_chain(x1, wa, b, False, x2=x2, w2=wb, b2=b, x3=x3)))
    xf = x1.permute(0, 2, 3, 1).reshape(-1, 128); bh = b.half()
    print(name, 'library linear+relu (one link) %.1f us' % t(lambda: torch.relu_(F.linear(xf, w1, bh))))
