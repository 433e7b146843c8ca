"""Does a plain torch kernel (no code of this repo) return different bits when it shares the chip with a matrix-core
kernel inside a captured graph?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinteraction_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
Hi, Wi = 112, 200
x = (torch.randn(6, 256, Hi, Wi, device='cuda', generator=g) * 0.5).clamp_(min=0).half().contiguous(memory_format=torch.channels_last)
conv = torch.nn.Conv2d(256, 128, 3, padding=1).cuda().half()
packed = ops.pack_conv3x3(conv.weight, conv.bias)
u = torch.randn(3_000_000, device='cuda', generator=g)
w = torch.randn(3_000_000, device='cuda', generator=g).abs() + 0.5
mm_a = torch.randn(4096, 4096, device='cuda', generator=g).half()
def side():
    r = u
    for _ in range(8):
        r = r / w + u
    return r
def run(name, heavy):
    with torch.no_grad():
        ref = side().clone()
        heavy()
        torch.cuda.synchronize()
        ss = torch.cuda.Stream()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            main = torch.cuda.current_stream()
            ss.wait_stream(main)
            with torch.cuda.stream(ss):
                out = side()
            keep = heavy()
            main.wait_stream(ss)
        bad = 0
        for it in range(400):
            gr.replay()
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                if bad <= 2:
                    idx = (out != ref).nonzero().flatten()
                    print('   replay', it, 'elements differing', idx.numel(), 'first', idx[:6].tolist())
        print(name, ': glitched replays', bad, 'of 400')
run('torch div/add chain || this repo conv', lambda: ops.conv3x3(x, *packed))
run('torch div/add chain || torch.mm fp16 (hipBLASLt)', lambda: mm_a @ mm_a)
