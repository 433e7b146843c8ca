"""Graph-replay timing (no host overhead) of the fused local-window attention variants.
Usage: python tools/la_bench2.py [variant ...] ; env LA_SHAPE=img|bev"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops

variants = [int(a) for a in sys.argv[1:]] or [5, 3, 4, 6]
n, C, H, W = (int(os.environ.get('LA_N', '6')), 128, 112, 200) if os.environ.get('LA_SHAPE', 'img') == 'img' else (1, 128, 180, 180)
g = torch.Generator(device='cuda').manual_seed(0)
mk = lambda: torch.randn(n, C, H, W, device='cuda', generator=g).relu().half().contiguous(memory_format=torch.channels_last)
sets = [(mk(), mk(), mk()) for _ in range(3)]          # rotate inputs: 3 x 138 MB > the 256 MB Infinity Cache
byt = 4 * n * C * H * W * 2
big = torch.randn(8192, 8192, device='cuda', dtype=torch.float16)
ref = None
for var in variants:
    f = lambda s: ops.local_attention(*s, 9, 9, 1 / math.sqrt(C), variant=var)
    out = f(sets[0]); torch.cuda.synchronize()
    if ref is None:
        ref = out.float()
    err = (out.float() - ref).abs().max().item()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for r in range(9):
            f(sets[r % 3])
    for _ in range(10):
        big @ big
    gr.replay(); gr.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        gr.replay()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 90 * 1e3
    print(f'variant {var}: {us:7.2f} us  {byt / us / 1e6:6.3f} TB/s ({byt / us / 1e6 / 8 * 100:4.1f}% of 8 TB/s)  max|diff vs first| {err:.2e}', flush=True)
