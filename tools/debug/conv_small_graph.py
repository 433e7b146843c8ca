"""Device time (hipGraph replay of 20 launches: no host launch overhead) of the class heat-map convolution 128 -> 10 on 180 x 180."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinteraction_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
for (Cin, Cout) in ((128, 10), (128, 128)):
    x = torch.randn(1, Cin, 180, 180, device='cuda', generator=g).relu().half().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).cuda().half()
    packed = ops.pack_conv3x3(conv.weight, conv.bias)
    f = lambda: ops.conv3x3(x, *packed)
    f(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    gr.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): gr.replay()
    e.record(); torch.cuda.synchronize()
    print(f'conv3x3 {Cin} -> {Cout} on 180 x 180: {s.elapsed_time(e) / 200 * 1e3:7.2f} us per launch (graph replay)')
