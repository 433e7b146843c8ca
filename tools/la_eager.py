"""Eager launches of window-attention variants on cold inputs (for rocprofv3 --pmc / --kernel-trace): python tools/la_eager.py v1 v2 ..."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops
n, C, H, W = (6, 128, 112, 200) if os.environ.get('LA_SHAPE', 'img') == 'img' else (1, 128, 180, 180)
g = torch.Generator(device='cuda').manual_seed(0)
mk = lambda: torch.randn(n, C, H, W, device='cuda', generator=g).relu().half().contiguous(memory_format=torch.channels_last)
sets = [(mk(), mk(), mk()) for _ in range(3)]
for var in [int(a) for a in sys.argv[1:]]:
    for r in range(9):
        ops.local_attention(*sets[r % 3], 9, 9, 1 / math.sqrt(C), variant=var)
    torch.cuda.synchronize()
