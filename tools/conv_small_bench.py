"""Graph-replay timing of the class head's last convolution (128 -> 10, 180 x 180).  DI_CONV_SMALL=0: conv3x3_kernel<4, 1, 1>."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops
x = [torch.randn(1, 128, 180, 180, device='cuda').relu().half().contiguous(memory_format=torch.channels_last) for _ in range(8)]
w = (torch.randn(10, 128, 3, 3, device='cuda') / 34).half()
b = torch.zeros(10, device='cuda').half()
packed = ops.pack_conv3x3(w, b)
f = lambda i: ops.conv3x3(x[i % 8], *packed, out_nchw=True, out_f32=True)
f(0); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(32):
        f(i)
g.replay(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); g.replay(); e.record(); torch.cuda.synchronize()
print(f'DI_CONV_SMALL={os.environ.get("DI_CONV_SMALL", "1")}: {s.elapsed_time(e) / 32 * 1e3:.2f} us per launch')
