import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinteraction_amd import ops
B, N, k = 1, 324000, 200
g0 = torch.Generator(device='cuda').manual_seed(0)
xs = []
for i in range(4):
    x = torch.rand(B, N, device='cuda', generator=g0)
    x[torch.rand(B, N, device='cuda', generator=g0) < 0.8] = 0
    xs.append(x)
static = xs[0].clone()
ref = [ops.topk(x, k).clone() for x in xs]
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    ops.topk(static, k)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    idx = ops.topk(static, k)
for it in range(8):
    static.copy_(xs[it % 4])
    gr.replay()
    torch.cuda.synchronize()
    print(it, 'identical', bool(torch.equal(idx, ref[it % 4])), 'overlap', len(set(idx[0].tolist()) & set(ref[it % 4][0].tolist())))
