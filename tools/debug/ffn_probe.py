"""FFN of the ++ layers at the image-token count: library GEMMs + element-wise vs the fused token kernels (run under
rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinteraction_amd import ops
M = 134400
g = torch.Generator(device='cuda').manual_seed(0)
x = (torch.randn(M, 128, device='cuda', generator=g) * 0.5).half()
w1 = (torch.randn(512, 128, device='cuda', generator=g) / 11).half(); b1 = torch.randn(512, device='cuda', generator=g) * 0.1
w2 = (torch.randn(128, 512, device='cuda', generator=g) / 22).half(); b2 = torch.randn(128, device='cuda', generator=g) * 0.1
lw, lb = torch.ones(128, device='cuda').half(), torch.zeros(128, device='cuda').half()
with torch.no_grad():
    for _ in range(5):
        h = torch.relu_(F.linear(x, w1, b1.half()))
        y = ops.add_layernorm(x, F.linear(h, w2, b2.half()), lw, lb, 1e-5)
        h2 = ops.token_linear(x, w1, b1, act1=1)
        y2 = ops.token_linear(h2, w2, b2, res1=x, ln1=(lw, lb))
    torch.cuda.synchronize()
    print('max diff', (y.float() - y2.float()).abs().max().item())
