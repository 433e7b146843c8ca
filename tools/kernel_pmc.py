"""A few launches of the hot kernels at the benched shapes (for rocprofv3 --pmc passes; see tools/pmc_session.sh)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops
which = sys.argv[1].split(',') if len(sys.argv) > 1 else ['conv', 'la', 'pw']
g = torch.Generator(device='cuda').manual_seed(0)
def cl(*s): return (torch.randn(*s, device='cuda', generator=g) * 0.5).clamp_(min=0).half().contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    if 'conv' in which:
        for (n, Cin, H, W) in [(6, 256, 112, 200), (1, 512, 180, 180)]:
            x = cl(n, Cin, H, W)
            conv = torch.nn.Conv2d(Cin, 128, 3, padding=1).cuda().half()
            packed = ops.pack_conv3x3(conv.weight, conv.bias)
            for _ in range(3):
                ops.conv3x3(x, *packed)
                ops.conv3x3(x, *packed, use_staged=False)
    if 'la' in which:
        q, k, v = cl(6, 128, 112, 200), cl(6, 128, 112, 200), cl(6, 128, 112, 200)
        for _ in range(3):
            ops.local_attention(q, k, v, 9, 9, 1 / math.sqrt(128))
        q, k, v = cl(1, 128, 180, 180), cl(1, 128, 180, 180), cl(1, 128, 180, 180)
        for _ in range(3):
            ops.local_attention(q, k, v, 9, 9, 1 / math.sqrt(128))
    if 'pw' in which:
        x = cl(6, 128, 112, 200)
        w = (torch.randn(128, 128, device='cuda', generator=g) / 11).half()
        b = torch.randn(128, device='cuda', generator=g) * 0.1
        for _ in range(3):
            ops.pointwise_chain(x, w, b, True, w2=w, b2=b, relu2=True)
torch.cuda.synchronize()
