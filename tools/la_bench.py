"""Timing of the fused local-window attention variants on the image-side shape (6x112x200x128 fp16)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops  # noqa: E402


def main():
    variants = [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5, 6]
    iters = int(os.environ.get('LA_ITERS', '30'))
    n, C, H, W = 6, 128, 112, 200
    g = torch.Generator(device='cuda').manual_seed(0)
    q, k, v = (torch.randn(n, C, H, W, device='cuda', generator=g).relu().half()
               .contiguous(memory_format=torch.channels_last) for _ in range(3))
    byt = 4 * n * C * H * W * 2
    for var in variants:
        for _ in range(3):
            ops.local_attention(q, k, v, 9, 9, 1 / math.sqrt(C), variant=var)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            ops.local_attention(q, k, v, 9, 9, 1 / math.sqrt(C), variant=var)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / iters * 1e3
        print(f'variant {var}: {us:7.1f} us  {byt / us / 1e6:6.3f} TB/s ({byt / us / 1e6 / 8 * 100:4.1f}% of 8 TB/s)', flush=True)


if __name__ == '__main__':
    main()
