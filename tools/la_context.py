"""Why is the merged window-attention launch 74.6 us inside the forward and 63.7 us alone?  The launch in different company
(one process per case, run under `rocprofv3 --kernel-trace --stats`: tools/la_context.sh reads the ring kernel's average).
Cases:  alone | fill (a 258 MB fill of an unrelated buffer before every launch) | produce (the launch's own q, k, v are
rewritten right before it: three 69 MB device copies) | gemm (a 4096^3 fp16 product before every launch: matrix cores, 32 MB out)
| relu (inputs with the sparsity of the forward's maps: half zeros)"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops

case = sys.argv[1] if len(sys.argv) > 1 else 'alone'
n, C, H, W = 12, 128, 112, 200
g = torch.Generator(device='cuda').manual_seed(0)
mk = lambda: torch.randn(n, C, H, W, device='cuda', generator=g).relu().half().contiguous(memory_format=torch.channels_last)
sets = [(mk(), mk(), mk()) for _ in range(3)]
src = (mk(), mk(), mk())
junk = torch.empty(258 * 1024 * 1024 // 2, device='cuda', dtype=torch.float16)
big = torch.randn(4096, 4096, device='cuda', dtype=torch.float16)
f = lambda s: ops.local_attention(*s, 9, 9, 1 / math.sqrt(C))
f(sets[0]); big @ big; torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for r in range(9):
        s = sets[r % 3]
        if case == 'fill':
            junk.fill_(1.0)
        elif case == 'produce':
            for d, x in zip(s, src):
                d.copy_(x)
        elif case == 'gemm':
            big @ big
        f(s)
for _ in range(12):
    gr.replay()
torch.cuda.synchronize()
print(case, 'done')
