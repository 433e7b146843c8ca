"""What bounds the window-attention launch: (i) streaming floor of its HBM bytes on this box (3 reads + 1 write of the map
size, an element-wise kernel), (ii) the kernel generations.  Inputs rotate over 3 sets (> Infinity Cache)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops

n, C, H, W = 6, 128, 112, 200
g = torch.Generator(device='cuda').manual_seed(0)
mk = lambda: torch.randn(n, C, H, W, device='cuda', generator=g).relu().half().contiguous(memory_format=torch.channels_last)
sets = [(mk(), mk(), mk()) for _ in range(3)]
outs = [torch.empty_like(sets[0][0]) for _ in range(3)]
byt = 4 * n * C * H * W * 2


def timed(name, f, b=byt):
    f(0); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for r in range(9):
            f(r % 3)
    gr.replay(); gr.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        gr.replay()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 90 * 1e3
    print(f'{name:58s} {us:7.2f} us  {b / us / 1e6:6.3f} TB/s', flush=True)


timed('addcmul (3 reads + 1 write, element-wise)', lambda i: torch.addcmul(sets[i][0], sets[i][1], sets[i][2], out=outs[i]))
timed('copy_ (1 read + 1 write)', lambda i: outs[i].copy_(sets[i][0]), byt // 2)
timed('add (2 reads + 1 write)', lambda i: torch.add(sets[i][0], sets[i][1], out=outs[i]), byt * 3 // 4)
for var, name in ((4, 'window attention, 2nd generation, 8x8 tiles, 2 workgroups per CU'), (16, 'LDS-DMA generation, 3 workgroups per CU'),
                  (17, 'LDS-DMA generation, 2 workgroups per CU'), (18, 'producer / consumer generation, 2 workgroups per CU')):
    timed(name, lambda i, var=var: ops.local_attention(*sets[i], 9, 9, 1 / math.sqrt(C), variant=var))
